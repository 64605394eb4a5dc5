/*
 * fake_rccl_files.c -- TEST INFRASTRUCTURE: a stand-in for librccl.so that implements the six entry points the product
 * resolves with dlsym (csrc/at_comm.hip: ncclGetUniqueId, ncclCommInitRank, ncclBroadcast, ncclAllGather, ncclCommDestroy,
 * ncclGetErrorString) over FILES in the job's rendezvous directory (AT_COMM_DIR, which the launcher of `alignTools batch
 * --gpus N` exports).  RCCL refuses two ranks on one device, and the test box has one card: with AT_RCCL_LIB pointing here
 * the N > 1 logic of the host -- shares of the pairs, the scoring broadcast, the sizes-then-payload gather, rank 0 printing
 * -- runs with several ranks on that card, through exactly the calls a node of 8 GPUs makes.  Built by tests/conftest.py into
 * a temporary directory; never part of libaligntools_hip.so.  The buffers are device pointers, as with RCCL.
 */
#define _POSIX_C_SOURCE 200809L
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct { char internal[128]; } fake_uid;
typedef struct { int rank, world; long seq; char dir[512]; } fake_comm;

static int put_file(const char *path, const void *p, size_t n)
{
	char tmp[640];
	FILE *f;
	snprintf(tmp, sizeof tmp, "%s.tmp", path);
	f = fopen(tmp, "wb");
	if (!f) return 1;
	if (n && fwrite(p, 1, n, f) != n) { fclose(f); return 1; }
	fclose(f);
	return rename(tmp, path) != 0;
}

static int get_file(const char *path, void *p, size_t n)
{
	int tries;
	for (tries = 0; tries < 300000; ++tries) {
		FILE *f = fopen(path, "rb");
		if (f) {
			const size_t got = n ? fread(p, 1, n, f) : 0;
			fclose(f);
			if (got == n) return 0;
		}
		usleep(2000);
	}
	return 1;
}

static size_t width(int dtype) { return dtype == 2 ? 4 : 1; }   /* ncclInt32 = 2, ncclInt8 = 0 */

int ncclGetUniqueId(fake_uid *id) { memset(id, 0x5a, sizeof *id); return 0; }

int ncclCommInitRank(void **comm, int nranks, fake_uid id, int rank)
{
	fake_comm *c = (fake_comm *)calloc(1, sizeof *c);
	const char *dir = getenv("AT_COMM_DIR");
	(void)id;
	if (!c || !dir) return 1;
	c->rank = rank; c->world = nranks;
	snprintf(c->dir, sizeof c->dir, "%s", dir);
	*comm = c;
	return 0;
}

int ncclBroadcast(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t stream)
{
	fake_comm *c = (fake_comm *)comm;
	const size_t n = count * width(dtype);
	char path[640], *host = (char *)malloc(n ? n : 1);
	int bad = 0;
	snprintf(path, sizeof path, "%s/fb%ld", c->dir, c->seq++);
	if (hipStreamSynchronize(stream) != hipSuccess) bad = 1;
	if (!bad && c->rank == root) {
		bad = hipMemcpy(host, send, n, hipMemcpyDeviceToHost) != hipSuccess || put_file(path, host, n);
		if (!bad && recv != send) bad = hipMemcpy(recv, host, n, hipMemcpyHostToDevice) != hipSuccess;
	} else if (!bad) {
		bad = get_file(path, host, n) || hipMemcpy(recv, host, n, hipMemcpyHostToDevice) != hipSuccess;
	}
	free(host);
	return bad;
}

int ncclAllGather(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t stream)
{
	fake_comm *c = (fake_comm *)comm;
	const size_t n = sendcount * width(dtype);
	char path[640], *host = (char *)malloc(n ? n : 1);
	const long q = c->seq++;
	int r, bad = 0;
	if (hipStreamSynchronize(stream) != hipSuccess) bad = 1;
	snprintf(path, sizeof path, "%s/fg%ld_%d", c->dir, q, c->rank);
	if (!bad) bad = hipMemcpy(host, send, n, hipMemcpyDeviceToHost) != hipSuccess || put_file(path, host, n);
	for (r = 0; !bad && r < c->world; ++r) {
		snprintf(path, sizeof path, "%s/fg%ld_%d", c->dir, q, r);
		bad = get_file(path, host, n) || hipMemcpy((char *)recv + (size_t)r * n, host, n, hipMemcpyHostToDevice) != hipSuccess;
	}
	free(host);
	return bad;
}

int ncclCommDestroy(void *comm) { free(comm); return 0; }
const char *ncclGetErrorString(int e) { return e ? "fake rccl (files): a collective failed" : "no error"; }
