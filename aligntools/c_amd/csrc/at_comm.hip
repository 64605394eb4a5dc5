/*
 * at_comm.hip -- the collectives of a multi-process batch (one process per GPU): RCCL broadcast of
 * the scoring block from rank 0 and an all-gather of every rank's results, behind the C ABI (at_comm_* in
 * include/aligntools_hip.h) so that the C host needs neither HIP nor RCCL headers.
 *
 * RCCL is loaded with dlopen when a communicator is created, not linked: the single-GPU paths (and processes that
 * bring their own RCCL, like PyTorch) never see it.  The rendezvous is a directory the launcher creates (AT_COMM_DIR):
 * rank 0 writes the ncclUniqueId there, the other ranks wait for it.
 *
 * AT_RCCL_LIB names another library to load in RCCL's place (default: librccl.so.1, then librccl.so).  The tests use it to
 * rehearse the N > 1 logic of the host with several ranks on ONE card -- RCCL refuses two ranks on one device -- against a
 * stand-in that implements the five entry points over files (tests/c/fake_rccl_files.c, built by the tests, never shipped in
 * this library: round 3 had that transport compiled in here as AT_COMM=files).
 */
#include "../../../include/aligntools_hip.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

extern "C" int at_handle_device(const at_handle *h);                    /* at_hip.hip */
extern "C" int at_comm_fail(at_handle *h, int code, const char *msg);   /* sets at_last_error */
extern "C" void at_get_scoring(const at_handle *h, int *v7, const int **sites);
extern "C" void **at_comm_slot(at_handle *h);                           /* where the handle keeps its communicator */

namespace {

typedef struct { char internal[128]; } uid_t128;
typedef int (*fn_uid)(uid_t128 *);
typedef int (*fn_init)(void **, int, uid_t128, int);
typedef int (*fn_bcast)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_errstr)(int);
constexpr int kNcclInt8 = 0, kNcclInt32 = 2;   /* ncclDataType_t values passed through the int-typed pointers above */

/* The declarations above are written by hand so that the build needs no RCCL.  Where the installed header exists (this
 * image: /opt/rocm/include/rccl/rccl.h) the compiler checks them against it: the size of the id passed BY VALUE, the two
 * enum values, and that every entry point has the argument list assumed here (ncclResult_t is an int-sized enum,
 * ncclComm_t a pointer: the int / void * spellings above are ABI-identical on x86-64). */
#if __has_include(<rccl/rccl.h>)
} /* namespace */
#include <rccl/rccl.h>
#include <type_traits>
namespace {
#define AT_RCCL_ABI_CHECKED 1
static_assert(sizeof(ncclUniqueId) == sizeof(uid_t128) && alignof(ncclUniqueId) == alignof(uid_t128), "ncclUniqueId is not 128 opaque bytes");
static_assert((int)ncclInt8 == kNcclInt8 && (int)ncclInt32 == kNcclInt32, "ncclDataType_t values moved");
static_assert((int)ncclSuccess == 0, "ncclSuccess is not 0");
static_assert(sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclComm_t) == sizeof(void *), "RCCL scalar types");
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId *)>::value, "ncclGetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t *, int, ncclUniqueId, int)>::value, "ncclCommInitRank");
static_assert(std::is_same<decltype(&ncclBroadcast), ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "ncclBroadcast");
static_assert(std::is_same<decltype(&ncclAllGather), ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t)>::value, "ncclAllGather");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "ncclCommDestroy");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char *(*)(ncclResult_t)>::value, "ncclGetErrorString");
#else
#define AT_RCCL_ABI_CHECKED 0
#endif

struct Comm {
	int rank = 0, world = 1;
	std::string dir;
	void *lib = nullptr, *comm = nullptr;
	fn_bcast bcast = nullptr; fn_allgather allgather = nullptr; fn_destroy destroy = nullptr; fn_errstr errstr = nullptr;
	hipStream_t stream = nullptr;
	void *d_buf = nullptr; size_t d_bytes = 0;
};

bool wait_for_file(const std::string &path, std::vector<char> &data, double timeout_s)
{
	for (double waited = 0; waited < timeout_s; waited += 0.002) {
		FILE *f = fopen(path.c_str(), "rb");
		if (f) {
			fseek(f, 0, SEEK_END);
			const long n = ftell(f);
			if (n < 0) { fclose(f); return false; }
			fseek(f, 0, SEEK_SET);
			data.resize((size_t)n);
			const size_t got = n ? fread(data.data(), 1, (size_t)n, f) : 0;
			fclose(f);
			if ((long)got == n) return true;
		}
		usleep(2000);
	}
	return false;
}

bool write_file_atomically(const std::string &path, const void *p, size_t n)
{
	const std::string tmp = path + ".tmp";
	FILE *f = fopen(tmp.c_str(), "wb");
	if (!f) return false;
	const bool ok = n == 0 || fwrite(p, 1, n, f) == n;
	fclose(f);
	return ok && rename(tmp.c_str(), path.c_str()) == 0;
}

int dev_buf(at_handle *h, Comm *c, size_t need)
{
	if (need <= c->d_bytes) return AT_OK;
	if (c->d_buf) (void)hipFree(c->d_buf);
	c->d_buf = nullptr; c->d_bytes = 0;
	if (hipMalloc(&c->d_buf, need + 4096) != hipSuccess) return at_comm_fail(h, AT_ERR_NOMEM, "at_comm: hipMalloc of the staging buffer failed");
	c->d_bytes = need + 4096;
	return AT_OK;
}

/* every rank contributes `n` bytes (the same n everywhere); `all` receives world * n bytes in rank order */
int allgather_fixed(at_handle *h, Comm *c, const void *mine, size_t n, void *all)
{
	if (c->world == 1 && !c->comm) { memcpy(all, mine, n); return AT_OK; }
	const size_t pad = (n + 15) & ~(size_t)15;
	int rc = dev_buf(h, c, pad * ((size_t)c->world + 1));
	if (rc) return rc;
	char *d_send = (char *)c->d_buf, *d_recv = d_send + pad;
	if (hipMemcpyAsync(d_send, mine, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) return at_comm_fail(h, AT_ERR_NODEVICE, "at_comm: upload failed");
	const int e = c->allgather(d_send, d_recv, pad, kNcclInt8, c->comm, c->stream);
	if (e) return at_comm_fail(h, AT_ERR_NODEVICE, c->errstr ? c->errstr(e) : "ncclAllGather failed");
	std::vector<char> host(pad * (size_t)c->world);
	if (hipMemcpyAsync(host.data(), d_recv, host.size(), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
		return at_comm_fail(h, AT_ERR_NODEVICE, "at_comm: download failed");
	for (int r = 0; r < c->world; ++r) memcpy((char *)all + (size_t)r * n, host.data() + (size_t)r * pad, n);
	return AT_OK;
}

} /* namespace */

static int comm_init(at_handle *h, int rank, int world, const char *dir)
{
	if (!h || world < 1 || rank < 0 || rank >= world || !dir) return at_comm_fail(h, AT_ERR_ARG, "at_comm_init: bad rank / world / directory");
	if (*at_comm_slot(h)) return at_comm_fail(h, AT_ERR_ARG, "at_comm_init: this handle already has a communicator (at_comm_destroy first)");
	Comm *c = new Comm();
	c->rank = rank; c->world = world; c->dir = dir;
	*at_comm_slot(h) = c;
	/* (a world of one needs no collective; AT_COMM_FORCE_RCCL=1 builds the communicator anyway, so that the RCCL branch --
	 * dlopen, ncclCommInitRank, ncclBroadcast, ncclAllGather -- can be exercised on a one-GPU box) */
	if (world == 1 && !getenv("AT_COMM_FORCE_RCCL")) return AT_OK;
	if (hipSetDevice(at_handle_device(h)) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
		return at_comm_fail(h, AT_ERR_NODEVICE, "at_comm_init: no stream on this rank's device");
	const char *named = getenv("AT_RCCL_LIB");
	if (named && *named) {
		c->lib = dlopen(named, RTLD_NOW | RTLD_LOCAL);
		if (!c->lib) return at_comm_fail(h, AT_ERR_NODEVICE, "at_comm_init: cannot load the library AT_RCCL_LIB names");
	} else {
		c->lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
		if (!c->lib) c->lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
		if (!c->lib) return at_comm_fail(h, AT_ERR_NODEVICE, "at_comm_init: librccl.so not found (needed for --gpus N > 1)");
	}
	fn_uid get_uid = (fn_uid)dlsym(c->lib, "ncclGetUniqueId");
	fn_init init = (fn_init)dlsym(c->lib, "ncclCommInitRank");
	c->bcast = (fn_bcast)dlsym(c->lib, "ncclBroadcast");
	c->allgather = (fn_allgather)dlsym(c->lib, "ncclAllGather");
	c->destroy = (fn_destroy)dlsym(c->lib, "ncclCommDestroy");
	c->errstr = (fn_errstr)dlsym(c->lib, "ncclGetErrorString");
	if (!get_uid || !init || !c->bcast || !c->allgather || !c->destroy) return at_comm_fail(h, AT_ERR_NODEVICE, "at_comm_init: librccl.so lacks a collective entry point");
	uid_t128 id;
	const std::string idfile = c->dir + "/rccl_unique_id";
	if (rank == 0) {
		const int e = get_uid(&id);
		if (e) return at_comm_fail(h, AT_ERR_NODEVICE, c->errstr ? c->errstr(e) : "ncclGetUniqueId failed");
		if (!write_file_atomically(idfile, &id, sizeof id)) return at_comm_fail(h, AT_ERR_ARG, "at_comm_init: cannot write the rendezvous file");
	} else {
		std::vector<char> d;
		if (!wait_for_file(idfile, d, 120.0) || d.size() != sizeof id) return at_comm_fail(h, AT_ERR_NODEVICE, "at_comm_init: rank 0 never published its RCCL id");
		memcpy(&id, d.data(), sizeof id);
	}
	/* RCCL prints a version banner on stdout when a communicator comes up: the CLI's stdout is its result, so the banner
	 * goes to stderr */
	fflush(stdout);
	const int saved = dup(1);
	if (saved >= 0) (void)dup2(2, 1);
	const int e = init(&c->comm, world, id, rank);
	fflush(stdout);
	if (saved >= 0) { (void)dup2(saved, 1); close(saved); }
	if (e) return at_comm_fail(h, AT_ERR_NODEVICE, c->errstr ? c->errstr(e) : "ncclCommInitRank failed");
	return AT_OK;
}

/* rank 0's scoring block (m, u, o, e, j, use_jump, sites) becomes every rank's: two broadcasts, the fixed part with the number
 * of sites first, then exactly that many sites */
static int comm_broadcast_scoring(at_handle *h)
{
	Comm *c = h ? (Comm *)*at_comm_slot(h) : nullptr;
	if (!c) return at_comm_fail(h, AT_ERR_ARG, "at_comm_broadcast_scoring: no communicator");
	int v[8] = {0};
	const int *sites = nullptr;
	at_get_scoring(h, v, &sites);                 /* v[0..5] = m,u,o,e,j,use_jump, v[6] = nsites */
	std::vector<int> st(sites, sites + v[6]);
	if (c->world > 1 || c->comm) {
		{
			int rc = dev_buf(h, c, 64);
			if (rc) return rc;
			if (hipMemcpyAsync(c->d_buf, v, 32, hipMemcpyHostToDevice, c->stream) != hipSuccess) return at_comm_fail(h, AT_ERR_NODEVICE, "at_comm: upload failed");
			int e = c->bcast(c->d_buf, c->d_buf, 8, kNcclInt32, 0, c->comm, c->stream);
			if (e) return at_comm_fail(h, AT_ERR_NODEVICE, c->errstr ? c->errstr(e) : "ncclBroadcast failed");
			if (hipMemcpyAsync(v, c->d_buf, 32, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
				return at_comm_fail(h, AT_ERR_NODEVICE, "at_comm: download failed");
			if (v[6] < 0) return at_comm_fail(h, AT_ERR_ARG, "at_comm: negative site count in the scoring block");
			if (v[6] > 0) {
				st.resize((size_t)v[6]);
				rc = dev_buf(h, c, (size_t)v[6] * 4);
				if (rc) return rc;
				if (hipMemcpyAsync(c->d_buf, st.data(), (size_t)v[6] * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) return at_comm_fail(h, AT_ERR_NODEVICE, "at_comm: upload failed");
				e = c->bcast(c->d_buf, c->d_buf, (size_t)v[6], kNcclInt32, 0, c->comm, c->stream);
				if (e) return at_comm_fail(h, AT_ERR_NODEVICE, c->errstr ? c->errstr(e) : "ncclBroadcast failed");
				if (hipMemcpyAsync(st.data(), c->d_buf, (size_t)v[6] * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
					return at_comm_fail(h, AT_ERR_NODEVICE, "at_comm: download failed");
			}
		}
	}
	return at_set_scoring(h, v[0], v[1], v[2], v[3], v[4], v[5], st.data(), v[6]);
}

/* gather with a different size per rank (sizes first, then one payload padded to the largest): `all` receives the parts
 * back to back in rank order, bytes_of_rank[r] (written by the call) says how long each is */
static int comm_allgather(at_handle *h, const void *mine, int64_t mine_bytes, void *all, int64_t all_cap, int64_t *bytes_of_rank)
{
	Comm *c = h ? (Comm *)*at_comm_slot(h) : nullptr;
	if (!c || mine_bytes < 0 || !bytes_of_rank || (mine_bytes && !mine)) return at_comm_fail(h, AT_ERR_ARG, "at_comm_allgather: bad argument");
	int rc = allgather_fixed(h, c, &mine_bytes, sizeof mine_bytes, bytes_of_rank);
	if (rc) return rc;
	int64_t mx = 0, tot = 0;
	for (int r = 0; r < c->world; ++r) { mx = std::max(mx, bytes_of_rank[r]); tot += bytes_of_rank[r]; }
	if (tot > all_cap || (tot && !all)) return at_comm_fail(h, AT_ERR_ARG, "at_comm_allgather: receive buffer too small");
	if (mx == 0) return AT_OK;
	std::vector<char> send((size_t)mx, 0), recv((size_t)mx * (size_t)c->world);
	if (mine_bytes) memcpy(send.data(), mine, (size_t)mine_bytes);
	rc = allgather_fixed(h, c, send.data(), (size_t)mx, recv.data());
	if (rc) return rc;
	int64_t o = 0;
	for (int r = 0; r < c->world; ++r) { memcpy((char *)all + o, recv.data() + (size_t)r * (size_t)mx, (size_t)bytes_of_rank[r]); o += bytes_of_rank[r]; }
	return AT_OK;
}

/* the C ABI: no C++ exception (std::bad_alloc of a staging vector ...) leaves the library */
#define AT_COMM_GUARD(call)                                                                          \
	try { return call; }                                                                             \
	catch (const std::exception &ex) { return at_comm_fail(h, AT_ERR_NOMEM, ex.what()); }           \
	catch (...) { return at_comm_fail(h, AT_ERR_NOMEM, "at_comm: unknown C++ exception"); }
extern "C" int at_comm_init(at_handle *h, int rank, int world, const char *dir) { AT_COMM_GUARD(comm_init(h, rank, world, dir)) }
extern "C" int at_comm_broadcast_scoring(at_handle *h) { AT_COMM_GUARD(comm_broadcast_scoring(h)) }
extern "C" int at_comm_allgather(at_handle *h, const void *mine, int64_t mine_bytes, void *all, int64_t all_cap, int64_t *bytes_of_rank)
{
	AT_COMM_GUARD(comm_allgather(h, mine, mine_bytes, all, all_cap, bytes_of_rank))
}
extern "C" int at_comm_abi_checked(void) { return AT_RCCL_ABI_CHECKED; }   /* 1: built against the installed rccl.h (static_asserts above) */

extern "C" void at_comm_destroy(at_handle *h)
{
	if (!h) return;
	Comm *c = (Comm *)*at_comm_slot(h);
	if (!c) return;
	if (c->comm && c->destroy) (void)c->destroy(c->comm);
	if (c->d_buf) (void)hipFree(c->d_buf);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	/* (the RCCL library stays loaded: its own threads may still be winding down) */
	delete c;
	*at_comm_slot(h) = nullptr;
}
