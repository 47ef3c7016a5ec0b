/* tests/fake_rccl.c -- TEST INFRASTRUCTURE: a stand-in for librccl that lets SEVERAL ranks share ONE GPU.
 *
 * RCCL refuses two ranks on the same device ("Duplicate GPU detected"), and the GPU boxes the tests run on have one
 * MI355X.  librxgpu binds its RCCL at run time ($RXGPU_RCCL_LIB, rxgpu_comm.c), so the tests point it at this file
 * instead and run the PRODUCT's multi-rank path -- rxgpu_comm_create on every rank, rxgpu_shard_tunes,
 * rxgpu_power_scan_run_sharded with its padded rows and its grouped gather, the root's merge and CSV -- with 2, 4 and
 * 8 processes on the one device.  Only the transport is fake: a gather is "wait for the stream, copy the send buffer to
 * a file in $FAKE_RCCL_DIR, the root reads every rank's file into its receive buffer".  Nothing here is DSP, nothing
 * here ships; the real RCCL runs with world 1 in test_gpu_power.py and with world N wherever N GPUs are visible.
 *
 * build: gcc -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.c -o libfake_rccl.so -L/opt/rocm/lib -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt32 = 2, ncclInt64 = 4 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct fake_comm { int rank, world; long seq; int group_depth; char dir[96]; char tag[24]; };
typedef struct fake_comm *ncclComm_t;

static const char *fake_dir(void)
{
	const char *d = getenv("FAKE_RCCL_DIR");
	return d && *d ? d : "/tmp";
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
	memset(id, 0, sizeof(*id));
	snprintf(id->internal, sizeof(id->internal), "fk%ld_%d", (long)getpid(), rand() & 0xffff);
	return ncclSuccess;
}

static int wait_file(const char *path, double seconds)
{
	for (int i = 0; i < (int)(seconds * 200); i++) {
		if (access(path, R_OK) == 0)
			return 0;
		usleep(5000);
	}
	return -1;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
	struct fake_comm *c = calloc(1, sizeof(*c));
	char path[256];
	if (!c || nranks < 1 || rank < 0 || rank >= nranks)
		return ncclInvalidArgument;
	c->rank = rank;
	c->world = nranks;
	snprintf(c->dir, sizeof(c->dir), "%s", fake_dir());
	snprintf(c->tag, sizeof(c->tag), "%.23s", id.internal);
	/* rendezvous: everybody announces itself, everybody waits for everybody */
	snprintf(path, sizeof(path), "%s/%s.join.%d", c->dir, c->tag, rank);
	FILE *f = fopen(path, "wb");
	if (!f)
		return ncclSystemError;
	fclose(f);
	for (int r = 0; r < nranks; r++) {
		snprintf(path, sizeof(path), "%s/%s.join.%d", c->dir, c->tag, r);
		if (wait_file(path, 120.0))
			return ncclSystemError;
	}
	*out = c;
	return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) { free(c); return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { *n = c->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) { *r = c->rank; return ncclSuccess; }
/* grouping changes nothing for this transport (every gather completes before it returns); the calls must pair up */
static int g_depth;
ncclResult_t ncclGroupStart(void) { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { if (g_depth <= 0) return ncclInvalidArgument; g_depth--; return ncclSuccess; }

ncclResult_t ncclGather(const void *send, void *recv, size_t count, ncclDataType_t dt, int root, ncclComm_t c, hipStream_t st)
{
	const size_t bytes = count * (dt == ncclInt64 ? 8 : 4);
	char path[256], tmp[272];
	void *host = malloc(bytes ? bytes : 1);
	if (!host || (dt != ncclInt64 && dt != ncclInt32))
		return ncclInvalidArgument;
	const long seq = c->seq++;
	/* everything enqueued on the stream before the gather (the scan) has to be done, as for the real collective */
	if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(host, send, bytes, hipMemcpyDeviceToHost) != hipSuccess)
		return ncclSystemError;
	snprintf(path, sizeof(path), "%s/%s.g%ld.%d", c->dir, c->tag, seq, c->rank);
	snprintf(tmp, sizeof(tmp), "%s.part", path);
	FILE *f = fopen(tmp, "wb");
	if (!f || fwrite(host, 1, bytes, f) != bytes)
		return ncclSystemError;
	fclose(f);
	if (rename(tmp, path))
		return ncclSystemError;
	if (c->rank == root) {
		for (int r = 0; r < c->world; r++) {
			snprintf(path, sizeof(path), "%s/%s.g%ld.%d", c->dir, c->tag, seq, r);
			if (wait_file(path, 120.0))
				return ncclSystemError;
			f = fopen(path, "rb");
			if (!f || fread(host, 1, bytes, f) != bytes)
				return ncclSystemError;
			fclose(f);
			unlink(path);
			if (hipMemcpy((char *)recv + (size_t)r * bytes, host, bytes, hipMemcpyHostToDevice) != hipSuccess)
				return ncclSystemError;
		}
	}
	free(host);
	return ncclSuccess;
}
