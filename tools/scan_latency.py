"""tools/scan_latency.py -- rxgpu_scan on the BASELINE configs[2] sweep (599 tunes x 16384 int16, separate caller buffers) in deferred mode: microseconds per sweep and per
interval sync with the host-clock phase table (rxgpu_scan_timing), the gather kernel's own duration (hip events), and for reference what one hipMemcpy of the same
19.6 MB from / to page-locked memory takes on this box (the PCIe bound of each)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RXGPU_DROPIN_TIMING"] = "1"
import numpy as np, torch
import rx_tools_amd as R
from rx_tools_amd.structs import TuningState
L = R.lib(); R.check(L.rxgpu_init(0))
plan = R.plan_range("24M:1.7G:1k", 0.0, 1)
n, T = 1 << plan.bin_e, plan.tune_count
wc, sw = R.window_coefs("rectangle", n), R.sine_table(plan.bin_e)
rng = np.random.default_rng(1)
mode = sys.argv[1] if len(sys.argv) > 1 else "separate"
if mode == "separate":                       # like frequency_range(): one malloc per tune for avg and buf16 (rtl_power.c:518-531)
    bufs = [rng.integers(-100, 101, plan.buf_len * 2, dtype=np.int16) for _ in range(T)]      # buf_len * 4 bytes allocated, buf_len int16 used
    avgs = [np.zeros(n, np.int64) for _ in range(T)]
else:                                        # rows of one array
    b2 = rng.integers(-100, 101, (T, plan.buf_len), dtype=np.int16); bufs = [b2[t] for t in range(T)]
    a2 = np.zeros((T, n), np.int64); avgs = [a2[t] for t in range(T)]
arr = (TuningState * T)()
for t in range(T):
    arr[t] = TuningState(plan.first_freq + t * plan.bw_seen, plan.rate, plan.bin_e, C.cast(avgs[t].ctypes.data, C.POINTER(C.c_int64)), 0, plan.downsample,
                         plan.downsample_passes, plan.crop, C.cast(bufs[t].ctypes.data, C.POINTER(C.c_int16)), plan.buf_len)
R.check(L.rxgpu_scan_deferred(1))
for _ in range(3):
    R.check(L.rxgpu_scan(arr, T, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
R.check(L.rxgpu_scan_sync(arr, T))
ph = (C.c_double * 8)()
L.rxgpu_scan_timing(ph, 8)
L.rxgpu_prof_reset(); L.rxgpu_prof_enable(2)
sweeps, intervals = 20, 5
t_scan = t_sync = 0.0
for _ in range(intervals):
    t0 = time.perf_counter()
    for _ in range(sweeps):
        R.check(L.rxgpu_scan(arr, T, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
    t1 = time.perf_counter()
    R.check(L.rxgpu_scan_sync(arr, T))
    t2 = time.perf_counter()
    t_scan += t1 - t0; t_sync += t2 - t1
L.rxgpu_prof_enable(0)
L.rxgpu_scan_timing(ph, 8)
# the interval after a report: every row zeroed by csv_dbm and known to be (rxgpu_scan_rows_cleared): the merge writes without reading
t_cl = 0.0
for _ in range(intervals):
    for a in avgs:
        a[:] = 0
    assert L.rxgpu_scan_rows_cleared(arr, T) == T
    for _ in range(2):
        R.check(L.rxgpu_scan(arr, T, wc.ctypes.data, sw.ctypes.data, 1, 0, 0))
    R.check(L.rxgpu_sync())
    t1 = time.perf_counter()
    R.check(L.rxgpu_scan_sync(arr, T))
    t_cl += time.perf_counter() - t1
names = ["scan: table", "scan: gather launch", "scan: transforms enqueue", "scan: wait for gather", "sync: D2H wait", "sync: host merge"]
print("mode %s  zero_copy %d  merge in place %d" % (mode, L.rxgpu_scan_zero_copy(), L.rxgpu_scan_sync_in_place()))
print("rxgpu_scan      %.1f us per sweep   phases: %s" % (t_scan / (sweeps * intervals) * 1e6, ", ".join("%s %.1f" % (names[i], ph[i] / ph[6]) for i in range(4))))
print("rxgpu_scan_sync %.1f us per interval phases: %s" % (t_sync / intervals * 1e6, ", ".join("%s %.1f" % (names[i], ph[i] / ph[7]) for i in (4, 5))))
print("rxgpu_scan_sync %.1f us per interval when csv_dbm has cleared the rows (write-only merge)" % (t_cl / intervals * 1e6))
for nm in ("pw_zc_gather", "pw_fft", "pw_zc_merge"):
    ms, k = C.c_double(0), C.c_long(0)
    L.rxgpu_prof_get(nm.encode(), C.byref(ms), C.byref(k))
    if k.value:
        print("kernel %-14s %.1f us per launch (%d launches)" % (nm, ms.value / k.value * 1e3, k.value))
L.rxgpu_scan_deferred(0); L.rxgpu_scan_release()
# the PCIe bound of each direction on this box: one copy of the same bytes from / to page-locked memory
for nbytes, what in ((T * plan.buf_len * 2, "H2D 19.6 MB (a sweep's input)"), (T * n * 8, "D2H 19.6 MB (an interval's sums)")):
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory(); d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for _ in range(3): (d.copy_(h, non_blocking=True) if what[0] == "H" else h.copy_(d, non_blocking=True)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): (d.copy_(h, non_blocking=True) if what[0] == "H" else h.copy_(d, non_blocking=True))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%-36s %.1f us = %.1f GB/s" % (what, dt * 1e6, nbytes / dt / 1e9))
