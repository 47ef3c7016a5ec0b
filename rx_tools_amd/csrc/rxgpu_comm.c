/* rxgpu_comm.c -- multi-GPU rx_power: tunes sharded over the ranks of one node, one RCCL gather per report
 * interval to merge the rows on the rank that prints the CSV.
 *
 * scanner()'s tunes are independent units (rtl_power.c:679-771: each tuning_state has its own buf16, avg[] and
 * samples); nothing crosses tunes until main() prints the rows in tune order (rtl_power.c:1047-1050).  So rank r
 * scans the contiguous range [r*per, min(T,(r+1)*per)), per = ceil(T/W), and its [per][N] int64 avg block and
 * [per] int32 samples (padded to `per` rows so that the collective is fixed-size) go to the root with ncclGather,
 * enqueued on the library's own stream right behind the scan kernels -- no host synchronisation, no reduction
 * (the rows are disjoint), no ring: xGMI is point to point and a gather is W-1 direct transfers into the root.  The two
 * gathers share one ncclGroup: a single collective launch per interval.
 *
 * librccl is bound at run time (dlopen), so librxgpu.so itself keeps loading on hosts without RCCL; the rccl.h
 * declarations are used for the types only. */
#include "rxgpu_internal.h"
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>

struct rxgpu_comm {
	ncclComm_t nccl;
	int rank, world;
	int owned;                       /* created here (destroy it) or adopted from the caller */
	long gathers;                    /* grouped gathers enqueued so far */
	/* rxgpu_power_scan_run_sharded: the gather of interval k runs on the copy stream while interval k + 1 is scanned (round 4) */
	hipEvent_t ev_scan;              /* the scan (and the padding memsets) of the interval whose gather is being enqueued */
	struct { const void *avg, *samples; hipEvent_t ev; int valid; } pend[2];   /* send buffers a gather may still be reading */
	int pend_next;
};

static struct {
	void *handle;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *);
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
	ncclResult_t (*CommDestroy)(ncclComm_t);
	ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
	ncclResult_t (*GroupStart)(void);
	ncclResult_t (*GroupEnd)(void);
	ncclResult_t (*CommCount)(const ncclComm_t, int *);
	ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
	const char *(*GetErrorString)(ncclResult_t);
	char path[256];
} g_rccl;
static pthread_mutex_t g_rccl_lock = PTHREAD_MUTEX_INITIALIZER;

static int rccl_bind(void)
{
	int rc = RXGPU_OK;
	pthread_mutex_lock(&g_rccl_lock);
	if (!g_rccl.handle) {
		/* $RXGPU_RCCL_LIB, else a librccl this process has already loaded (one RCCL per process: a host application
		 * that brought its own keeps using it), else the loader's search path, else ROCm's default location */
		const char *env = getenv("RXGPU_RCCL_LIB");
		const char *cand[] = { env, "librccl.so.1", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so" };
		const int flags[] = { RTLD_NOW | RTLD_LOCAL, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD, RTLD_NOW | RTLD_LOCAL, RTLD_NOW | RTLD_LOCAL, RTLD_NOW | RTLD_LOCAL };
		for (unsigned i = 0; i < sizeof(cand) / sizeof(cand[0]) && !g_rccl.handle; i++) {
			if (!cand[i] || !*cand[i])
				continue;
			g_rccl.handle = dlopen(cand[i], flags[i]);
			if (g_rccl.handle)
				snprintf(g_rccl.path, sizeof(g_rccl.path), "%s%s", cand[i], (flags[i] & RTLD_NOLOAD) ? " (already loaded)" : "");
		}
		if (!g_rccl.handle) {
			rc = rxgpu_fail(RXGPU_ENODEV, "librccl not found (set RXGPU_RCCL_LIB): %s", dlerror());
		} else {
#define BIND(field, sym) do { *(void **)&g_rccl.field = dlsym(g_rccl.handle, sym); if (!g_rccl.field && rc == RXGPU_OK) \
	rc = rxgpu_fail(RXGPU_ENODEV, "%s lacks %s", g_rccl.path, sym); } while (0)
			BIND(GetUniqueId, "ncclGetUniqueId");
			BIND(CommInitRank, "ncclCommInitRank");
			BIND(CommDestroy, "ncclCommDestroy");
			BIND(Gather, "ncclGather");
			BIND(GroupStart, "ncclGroupStart");
			BIND(GroupEnd, "ncclGroupEnd");
			BIND(CommCount, "ncclCommCount");
			BIND(CommUserRank, "ncclCommUserRank");
			BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
			if (rc != RXGPU_OK) {
				dlclose(g_rccl.handle);
				g_rccl.handle = NULL;
			}
		}
	}
	pthread_mutex_unlock(&g_rccl_lock);
	return rc;
}

#define RX_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) \
	return rxgpu_fail(RXGPU_ENODEV, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?", __FILE__, __LINE__); } while (0)

const char *rxgpu_comm_library(void)
{
	return rccl_bind() == RXGPU_OK ? g_rccl.path : NULL;
}

int rxgpu_comm_unique_id(void *id128)
{
	int rc;
	if (!id128)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_comm_unique_id: null buffer");
	if ((rc = rccl_bind()) != RXGPU_OK)
		return rc;
	ncclUniqueId id;
	RX_NCCL(g_rccl.GetUniqueId(&id));
	memcpy(id128, &id, sizeof(id));
	return RXGPU_OK;
}

/* what the communicator itself says about its size and this process's place in it must be what the caller sharded by */
static int comm_verify(const rxgpu_comm *c)
{
	int n = -1, r = -1;
	RX_NCCL(g_rccl.CommCount(c->nccl, &n));
	RX_NCCL(g_rccl.CommUserRank(c->nccl, &r));
	if (n != c->world || r != c->rank)
		return rxgpu_fail(RXGPU_EINVAL, "communicator is rank %d of %d, caller said rank %d of %d", r, n, c->rank, c->world);
	return RXGPU_OK;
}

int rxgpu_comm_create(rxgpu_comm **out, const void *id128, int rank, int world)
{
	int rc;
	if (!out || !id128 || world < 1 || rank < 0 || rank >= world)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_comm_create: bad arguments (rank %d of %d)", rank, world);
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)          /* the communicator binds to the device this process drives */
		return rc;
	if ((rc = rccl_bind()) != RXGPU_OK)
		return rc;
	rxgpu_comm *c = calloc(1, sizeof(*c));
	if (!c)
		return rxgpu_fail(RXGPU_ENOMEM, "out of host memory");
	ncclUniqueId id;
	memcpy(&id, id128, sizeof(id));
	ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
	if (r != ncclSuccess) {
		free(c);
		return rxgpu_fail(RXGPU_ENODEV, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
	}
	c->rank = rank;
	c->world = world;
	c->owned = 1;
	if ((rc = comm_verify(c)) != RXGPU_OK) {
		g_rccl.CommDestroy(c->nccl);
		free(c);
		return rc;
	}
	*out = c;
	return RXGPU_OK;
}

int rxgpu_comm_adopt(rxgpu_comm **out, void *nccl_comm, int rank, int world)
{
	int rc;
	if (!out || !nccl_comm || world < 1 || rank < 0 || rank >= world)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_comm_adopt: bad arguments");
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)          /* the gather goes on the library's stream: it has to exist */
		return rc;
	if ((rc = rccl_bind()) != RXGPU_OK)
		return rc;
	rxgpu_comm *c = calloc(1, sizeof(*c));
	if (!c)
		return rxgpu_fail(RXGPU_ENOMEM, "out of host memory");
	c->nccl = (ncclComm_t)nccl_comm;
	c->rank = rank;
	c->world = world;
	c->owned = 0;
	if ((rc = comm_verify(c)) != RXGPU_OK) {
		free(c);
		return rc;
	}
	*out = c;
	return RXGPU_OK;
}

void rxgpu_comm_destroy(rxgpu_comm *c)
{
	if (!c)
		return;
	if (rxgpu_hip_stream())
		hipStreamSynchronize(rxgpu_hip_stream());
	if (rxgpu_hip_stream3())
		hipStreamSynchronize(rxgpu_hip_stream3());
	if (c->owned && c->nccl && g_rccl.CommDestroy)
		g_rccl.CommDestroy(c->nccl);
	if (c->ev_scan)
		hipEventDestroy(c->ev_scan);
	for (int i = 0; i < 2; i++)
		if (c->pend[i].ev)
			hipEventDestroy(c->pend[i].ev);
	free(c);
}

int rxgpu_comm_rank(const rxgpu_comm *c) { return c ? c->rank : 0; }
int rxgpu_comm_world(const rxgpu_comm *c) { return c ? c->world : 1; }

/* contiguous tune ranges: rank r owns [first, first+count), every rank's block is padded to `per` rows */
int rxgpu_shard_tunes(int rank, int world, int total, int *first, int *count, int *per)
{
	if (world < 1 || rank < 0 || rank >= world || total < 0)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_shard_tunes: rank %d of %d, %d tunes", rank, world, total);
	const int p = (total + world - 1) / world;
	int lo = rank * p;
	if (lo > total) lo = total;
	int hi = lo + p;
	if (hi > total) hi = total;
	if (first) *first = lo;
	if (count) *count = hi - lo;
	if (per) *per = p;
	return RXGPU_OK;
}

/* ONE collective launch per report interval: the avg block and the samples vector of every rank travel in the same
 * ncclGroup (RCCL fuses the grouped gathers' point-to-point transfers into a single kernel on the stream), enqueued on the
 * library's stream right behind the scan -- rows merge where the reference prints them, rtl_power.c:1047-1050. */
static int gather_on(rxgpu_comm *c, hipStream_t st, const int64_t *d_avg_local, const int32_t *d_samples_local, int per, int n_bins,
                     int64_t *d_avg_all, int32_t *d_samples_all, int root)
{
	int rc;
	if (per < 0 || n_bins < 1)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_power_gather: bad arguments");
	const int world = c ? c->world : 1, rank = c ? c->rank : 0;
	if (root < 0 || root >= world)
		return rxgpu_fail(RXGPU_EINVAL, "root %d outside [0,%d)", root, world);
	if (per == 0)                                        /* a sweep of no tunes: nothing to merge, on every rank alike */
		return RXGPU_OK;
	if (!d_avg_local || !d_samples_local)
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_power_gather: null send buffer");
	if (rank == root && (!d_avg_all || !d_samples_all))
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_power_gather: the root needs the receive buffers");
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)
		return rc;
	if (!st)
		st = rxgpu_hip_stream();
	const size_t n_avg = (size_t)per * (size_t)n_bins;
	if (!c) {                                            /* one process: the "gather" is the rank's own block */
		if (d_avg_all != d_avg_local)
			RX_HIP(hipMemcpyAsync(d_avg_all, d_avg_local, n_avg * 8, hipMemcpyDeviceToDevice, st));
		if (d_samples_all != d_samples_local)
			RX_HIP(hipMemcpyAsync(d_samples_all, d_samples_local, (size_t)per * 4, hipMemcpyDeviceToDevice, st));
		return RXGPU_OK;
	}
	rxgpu_prof_begin_on("pw_gather", st);
	ncclResult_t r = g_rccl.GroupStart();
	if (r == ncclSuccess) {
		ncclResult_t r1 = g_rccl.Gather(d_avg_local, d_avg_all, n_avg, ncclInt64, root, c->nccl, st);
		ncclResult_t r2 = r1 == ncclSuccess ? g_rccl.Gather(d_samples_local, d_samples_all, (size_t)per, ncclInt32, root, c->nccl, st) : r1;
		r = g_rccl.GroupEnd();                           /* always closed, also after a failed enqueue */
		if (r2 != ncclSuccess)
			r = r2;
	}
	if (r != ncclSuccess) {
		rxgpu_prof_abort();
		return rxgpu_fail(RXGPU_ENODEV, "grouped ncclGather failed: %s", g_rccl.GetErrorString(r));
	}
	rxgpu_prof_end_on("pw_gather", st);
	c->gathers++;
	return RXGPU_OK;
}

/* the public form: on the library's stream, ordered with whatever the caller enqueues there next */
int rxgpu_power_gather(rxgpu_comm *c, const int64_t *d_avg_local, const int32_t *d_samples_local, int per, int n_bins,
                       int64_t *d_avg_all, int32_t *d_samples_all, int root)
{
	return gather_on(c, NULL, d_avg_local, d_samples_local, per, n_bins, d_avg_all, d_samples_all, root);
}

long rxgpu_comm_gathers(const rxgpu_comm *c) { return c ? c->gathers : 0; }

/* One report interval of a sharded sweep.  The scan goes on the library's stream, the grouped gather behind it on the COPY stream
 * (an event orders them): the next interval's scan does not wait for this interval's rows to cross xGMI -- at eight ranks the
 * root takes in 7 x 2.4 MB per interval while each rank scans 75 tunes.  What the gather still reads must not be scanned into:
 * a call whose send buffers are those of a gather that may be in flight (a caller that alternates two buffer sets meets its own
 * gather two intervals later, one that reuses a single set meets it at once) makes its scan wait for that gather.  The root's
 * receive buffers are filled in interval order (one stream); rxgpu_sync() covers both streams. */
int rxgpu_power_scan_run_sharded(rxgpu_power_scan *s, rxgpu_comm *c, const int16_t *d_in_local, int passes, int total_tunes,
                                 int64_t *d_avg_local, int32_t *d_samples_local, int n_bins,
                                 int64_t *d_avg_all, int32_t *d_samples_all, int root)
{
	int rc, count = 0, per = 0;
	if ((rc = rxgpu_shard_tunes(c ? c->rank : 0, c ? c->world : 1, total_tunes, NULL, &count, &per)) != RXGPU_OK)
		return rc;
	if (per > 0 && (!d_avg_local || !d_samples_local || n_bins < 1))
		return rxgpu_fail(RXGPU_EINVAL, "rxgpu_power_scan_run_sharded: null local buffers");
	if ((rc = rxgpu_ensure_init()) != RXGPU_OK)
		return rc;
	hipStream_t sa = rxgpu_hip_stream(), sc = rxgpu_hip_stream3();
	if (c) {
		for (int i = 0; i < 2; i++)
			if (c->pend[i].valid && (c->pend[i].avg == (const void *)d_avg_local || c->pend[i].samples == (const void *)d_samples_local)) {
				RX_HIP(hipStreamWaitEvent(sa, c->pend[i].ev, 0));
				c->pend[i].valid = 0;
			}
	}
	if (count > 0 && (rc = rxgpu_power_scan_run(s, d_in_local, passes, count, d_avg_local, d_samples_local)) != RXGPU_OK)
		return rc;
	if (count < per) {
		/* the padding rows of a short last rank (599 tunes over 8 ranks: 75 x 7 + 74) reach the root as zeros, whatever the
		 * caller left in them */
		RX_HIP(hipMemsetAsync(d_avg_local + (size_t)count * (size_t)n_bins, 0, (size_t)(per - count) * (size_t)n_bins * 8, sa));
		RX_HIP(hipMemsetAsync(d_samples_local + count, 0, (size_t)(per - count) * 4, sa));
	}
	if (!c || per == 0)
		return gather_on(c, sa, d_avg_local, d_samples_local, per, n_bins, d_avg_all, d_samples_all, root);
	if (!c->ev_scan)
		RX_HIP(hipEventCreateWithFlags(&c->ev_scan, hipEventDisableTiming));
	RX_HIP(hipEventRecord(c->ev_scan, sa));
	RX_HIP(hipStreamWaitEvent(sc, c->ev_scan, 0));
	if ((rc = gather_on(c, sc, d_avg_local, d_samples_local, per, n_bins, d_avg_all, d_samples_all, root)) != RXGPU_OK)
		return rc;
	const int k = c->pend_next;
	c->pend_next ^= 1;
	/* two entries: a caller rotating three or more buffer sets would evict a gather that may still be reading its send buffers --
	 * the scan stream then waits for the evicted gather before anything later can be scanned into those buffers */
	if (c->pend[k].valid)
		RX_HIP(hipStreamWaitEvent(sa, c->pend[k].ev, 0));
	if (!c->pend[k].ev)
		RX_HIP(hipEventCreateWithFlags(&c->pend[k].ev, hipEventDisableTiming));
	RX_HIP(hipEventRecord(c->pend[k].ev, sc));
	c->pend[k].avg = d_avg_local;
	c->pend[k].samples = d_samples_local;
	c->pend[k].valid = 1;
	return RXGPU_OK;
}
