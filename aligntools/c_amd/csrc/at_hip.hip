/*
 * at_hip.hip -- the C-ABI shim (include/aligntools_hip.h) over the gfx950
 * sweep kernel (at_sweep.hip.h).  Host C++ here is plumbing only: argument
 * checks, packing, the choice of storage class, launches.  There is no CPU
 * compute path: every DP cell is computed on the GPU or the call fails.
 */
#include "at_launch.h"
#include "at_pack.hip.h"
#include "at_render.hip.h"
#include "at_myers.hip.h"
#include "../../../include/aligntools_hip.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

using at::SweepArgs;
using at::Sweep16Args;

/* The host entry's helper threads: chunk c of a batch runs on worker c (chunk 0 on the caller's thread).  They live as long as the
 * handle -- starting five threads per call was a tenth of a millisecond before any byte moved. */
struct HostPool {
	std::vector<std::thread> th;
	std::mutex mu;
	std::condition_variable cv_go, cv_done;
	std::function<void(int)> job;
	unsigned long long gen = 0;
	int active = 0, pending = 0;
	bool stop = false;
	void worker(int w)
	{
		unsigned long long seen = 0;
		for (;;) {
			std::function<void(int)> fn;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_go.wait(lk, [&] { return stop || gen != seen; });
				if (stop) return;
				seen = gen;
				if (w >= active) continue;
				fn = job;
			}
			fn(w);
			{
				std::lock_guard<std::mutex> lk(mu);
				if (--pending == 0) cv_done.notify_all();
			}
		}
	}
	/* run fn(1) .. fn(n - 1) on the workers and fn(0) here; returns when all are done */
	void run(int n, const std::function<void(int)> &fn)
	{
		while ((int)th.size() < n - 1) { const int w = (int)th.size() + 1; th.emplace_back([this, w] { worker(w); }); }
		{
			std::lock_guard<std::mutex> lk(mu);
			job = fn; active = n; pending = n - 1; ++gen;
		}
		cv_go.notify_all();
		fn(0);
		std::unique_lock<std::mutex> lk(mu);
		cv_done.wait(lk, [&] { return pending == 0; });
	}
	~HostPool()
	{
		{ std::lock_guard<std::mutex> lk(mu); stop = true; }
		cv_go.notify_all();
		for (auto &t : th) t.join();
	}
};

struct at_handle {
	int device = 0;
	int ncu = 256;
	size_t lds_per_cu = 160 * 1024;
	hipStream_t stream = nullptr;   /* used by the host-buffer entry */
	/* scoring */
	int m = 1, u = -2, o = -5, e = -1, j = -10, use_jump = 0;
	std::vector<int> sites;
	int min_on = 0, min_score = 0;  /* at_set_min_score: all-vs-all overlap scores skip pairs proven below it */
	/* device scratch (grow-only) */
	uint32_t *d_sitemask = nullptr; size_t sitemask_words = 0; int sitemask_for_l2 = -1; bool sitemask_dirty = true;
	uint32_t *d_ws = nullptr; size_t ws_bytes = 0;
	uint32_t *d_ck = nullptr; size_t ck_bytes = 0;   /* two-pass tracebacks with a walk kernel: a launch's checkpoints */
	bool ck_alloc_failed = false;                    /* ... could not be allocated once: this handle keeps to the rounds inside the sweep's kernel */
	unsigned long long *d_queue = nullptr;
	int last_span = 0;              /* max_len1 + max_len2 of the handle's latest batch: the rendering kernel's hint for its group width */
	void *d_in = nullptr; size_t in_bytes = 0;
	void *d_out = nullptr; size_t out_bytes = 0;
	void *d_desc = nullptr; size_t desc_bytes = 0;
	void *d_order = nullptr; size_t order_bytes = 0;
	void *d_str = nullptr; size_t str_bytes = 0;
	void *d_scan = nullptr; size_t scan_bytes = 0;
	int *d_rflag = nullptr;         /* at_render_k's "op list walks off its sequences" flag */
	/* host entry: page-locked staging -- descriptor block, raw bytes on their way up, results and payload on their way down */
	void *hp_desc = nullptr; size_t hp_desc_bytes = 0;
	void *hp_blob = nullptr; size_t hp_blob_bytes = 0;
	void *hp_out = nullptr; size_t hp_out_bytes = 0;
	void *hp_flag = nullptr;
	void *hp_order = nullptr; size_t hp_order_bytes = 0;   /* the processing order of a ragged batch */
	double last_payload_per_pair = 16.0;   /* traceback bytes per pair of the latest batch: how much payload the next one fetches unasked */
	/* all-vs-all in slices: a copy stream, pinned result buffers (two sets, used in turn) and their events */
	hipStream_t copy_stream = nullptr;
	void *h_pin = nullptr; size_t pin_bytes = 0;
	hipEvent_t ev_sweep[2] = {nullptr, nullptr}, ev_copy[2] = {nullptr, nullptr};
	std::vector<at_handle *> kids;  /* helper handles of the host entry: chunks of one batch in flight side by side */
	HostPool *pool = nullptr;       /* ... and the threads they run on */
	void *comm = nullptr;           /* multi-process batches: the communicator (at_comm.hip) */
	char err[512] = {0};
	char cfg[320] = "none";
};

static thread_local char g_err[512] = "no error";   /* per thread: the host entry runs chunks on helper threads */

static int fail(at_handle *h, int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	snprintf(g_err, sizeof g_err, "%s", buf);
	if (h) snprintf(h->err, sizeof h->err, "%s", buf);
	return code;
}

/* no C++ exception crosses the C ABI (std::bad_alloc from a vector or a string on the host side): it becomes an error code */
template <typename F>
static int guarded(at_handle *h, const char *who, F &&body)
{
	try { return body(); }
	catch (const std::bad_alloc &) { return fail(h, AT_ERR_NOMEM, "%s: out of host memory", who); }
	catch (const std::exception &ex) { return fail(h, AT_ERR_NOMEM, "%s: %s", who, ex.what()); }
	catch (...) { return fail(h, AT_ERR_NOMEM, "%s: unknown C++ exception", who); }
}

#define HIP_TRY(h, call)                                                                      \
	do {                                                                                      \
		hipError_t e_ = (call);                                                               \
		if (e_ != hipSuccess)                                                                 \
			return fail((h), AT_ERR_NODEVICE, "%s: %s", #call, hipGetErrorString(e_));        \
	} while (0)

extern "C" const char *at_last_error(const at_handle *h) { return h ? h->err : g_err; }
extern "C" const char *at_last_config(const at_handle *h) { return h ? h->cfg : "none"; }
/* diagnostic builds (-DAT_TP_STATS=1): the 8 words of the handle's work counter block, after a device synchronisation */
extern "C" int at_debug_counters(at_handle *h, unsigned long long *out8)
{
	if (!h || !out8 || !h->d_queue) return AT_ERR_ARG;
	if (hipDeviceSynchronize() != hipSuccess) return AT_ERR_NODEVICE;
	return hipMemcpy(out8, h->d_queue, 64, hipMemcpyDeviceToHost) == hipSuccess ? AT_OK : AT_ERR_NODEVICE;
}

extern "C" int at_init(const int *device_ids, int n_devices, at_handle **out)
{
	if (!out) return fail(nullptr, AT_ERR_ARG, "at_init: out is NULL");
	*out = nullptr;
	if (n_devices != 1 && !(n_devices == 0 && !device_ids))
		return fail(nullptr, AT_ERR_ARG, "at_init: one process per GPU -- n_devices must be 1 (got %d)", n_devices);
	int count = 0;
	hipError_t e = hipGetDeviceCount(&count);
	if (e != hipSuccess || count <= 0)
		return fail(nullptr, AT_ERR_NODEVICE, "at_init: no HIP device (%s); there is no CPU fallback",
		            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
	int dev = 0;
	if (device_ids) dev = device_ids[0];
	else if (hipGetDevice(&dev) != hipSuccess) dev = 0;
	if (dev < 0 || dev >= count) return fail(nullptr, AT_ERR_ARG, "at_init: device %d out of range (0..%d)", dev, count - 1);
	at_handle *h = new at_handle();
	h->device = dev;
	/* a failing step frees what the earlier ones made: the caller gets no handle to destroy */
	auto init = [&]() -> int {
		HIP_TRY(h, hipSetDevice(dev));
		hipDeviceProp_t prop;
		HIP_TRY(h, hipGetDeviceProperties(&prop, dev));
		h->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
		if (prop.maxSharedMemoryPerMultiProcessor > 0) h->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
		HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
		HIP_TRY(h, hipMalloc((void **)&h->d_rflag, 256));
		HIP_TRY(h, hipMemset(h->d_rflag, 0, 256));
		HIP_TRY(h, hipHostMalloc(&h->hp_flag, 256, hipHostMallocDefault));
		return AT_OK;
	};
	const int rc = init();
	if (rc != AT_OK) { at_destroy(h); return rc; }   /* (the message is in the calling thread's at_last_error(NULL)) */
	h->err[0] = 0;
	*out = h;
	return AT_OK;
}

extern "C" void at_comm_destroy(at_handle *h);
/* what at_comm.hip needs of a handle */
extern "C" int at_handle_device(const at_handle *h) { return h->device; }
extern "C" int at_comm_fail(at_handle *h, int code, const char *msg) { return fail(h, code, "%s", msg); }
extern "C" void **at_comm_slot(at_handle *h) { return &h->comm; }
extern "C" void at_get_scoring(const at_handle *h, int *v7, const int **sites)
{
	v7[0] = h->m; v7[1] = h->u; v7[2] = h->o; v7[3] = h->e; v7[4] = h->j; v7[5] = h->use_jump; v7[6] = (int)h->sites.size();
	*sites = h->sites.data();
}

extern "C" void at_destroy(at_handle *h)
{
	if (!h) return;
	if (h->comm) at_comm_destroy(h);
	delete h->pool; h->pool = nullptr;      /* (joins its threads) */
	for (at_handle *k : h->kids) at_destroy(k);
	h->kids.clear();
	(void)hipSetDevice(h->device);
	if (h->d_sitemask) (void)hipFree(h->d_sitemask);
	if (h->d_ws) (void)hipFree(h->d_ws);
	if (h->d_ck) (void)hipFree(h->d_ck);
	if (h->d_queue) (void)hipFree(h->d_queue);
	if (h->d_in) (void)hipFree(h->d_in);
	if (h->d_out) (void)hipFree(h->d_out);
	if (h->d_desc) (void)hipFree(h->d_desc);
	if (h->d_order) (void)hipFree(h->d_order);
	if (h->d_str) (void)hipFree(h->d_str);
	if (h->d_scan) (void)hipFree(h->d_scan);
	if (h->d_rflag) (void)hipFree(h->d_rflag);
	if (h->h_pin) (void)hipHostFree(h->h_pin);
	if (h->hp_desc) (void)hipHostFree(h->hp_desc);
	if (h->hp_blob) (void)hipHostFree(h->hp_blob);
	if (h->hp_out) (void)hipHostFree(h->hp_out);
	if (h->hp_flag) (void)hipHostFree(h->hp_flag);
	if (h->hp_order) (void)hipHostFree(h->hp_order);
	for (int q = 0; q < 2; ++q) {
		if (h->ev_sweep[q]) (void)hipEventDestroy(h->ev_sweep[q]);
		if (h->ev_copy[q]) (void)hipEventDestroy(h->ev_copy[q]);
	}
	if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
	if (h->stream) (void)hipStreamDestroy(h->stream);
	delete h;
}

extern "C" int at_set_min_score(at_handle *h, int enabled, int32_t min_score)
{
	if (!h) return fail(nullptr, AT_ERR_ARG, "at_set_min_score: NULL handle");
	h->min_on = enabled ? 1 : 0;
	h->min_score = min_score;
	return AT_OK;
}

extern "C" int at_set_scoring(at_handle *h, int m, int u, int o, int e, int j, int use_jump, const int *sites, int nsites)
{
	if (!h) return fail(nullptr, AT_ERR_ARG, "at_set_scoring: NULL handle");
	if (nsites < 0 || (nsites > 0 && !sites)) return fail(h, AT_ERR_ARG, "at_set_scoring: bad site list");
	h->m = m; h->u = u; h->o = o; h->e = e; h->j = j; h->use_jump = use_jump ? 1 : 0;
	h->sites.assign(sites, sites + nsites);
	h->sitemask_dirty = true;
	return AT_OK;
}

/* ------------------------------------------------------------------ packing */

static inline int code2(uint8_t c)
{
	switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

extern "C" int64_t at_pack_words(int64_t npairs, const int32_t *len1, const int32_t *len2, int bits)
{
	if (npairs < 0 || !len1 || !len2 || (bits != 2 && bits != 8)) return -1;
	const int bpw = 32 / bits;
	int64_t w = 0;
	for (int64_t k = 0; k < npairs; ++k) {
		if (len1[k] < 0 || len2[k] < 0) return -1;
		w += (len1[k] + bpw - 1) / bpw + 1 + (len2[k] + bpw - 1) / bpw + 1;   /* +1: the window reads one word ahead */
	}
	return w + 4;
}

static void pack_one(const uint8_t *s, int len, int bits, uint32_t *dst)
{
	const int bpw = 32 / bits;
	const int nw = (len + bpw - 1) / bpw + 1;
	for (int w = 0; w < nw; ++w) {
		uint32_t v = 0;
		for (int b = 0; b < bpw; ++b) {
			const int idx = w * bpw + b;
			if (idx >= len) break;
			const uint32_t c = bits == 2 ? (uint32_t)code2(s[idx]) : (uint32_t)s[idx];
			v |= c << (b * bits);
		}
		dst[w] = v;
	}
}

extern "C" int at_pack_batch(int64_t npairs, const uint8_t *seq_blob,
                             const int64_t *off1, const int32_t *len1, const int64_t *off2, const int32_t *len2,
                             int bits, int *bits_out, uint32_t *words_out, int64_t *woff1_out, int64_t *woff2_out)
{
	if (npairs < 0 || !seq_blob || !off1 || !len1 || !off2 || !len2 || !woff1_out || !woff2_out)
		return fail(nullptr, AT_ERR_ARG, "at_pack_batch: NULL argument");
	if (bits == 0) {
		bits = 2;
		for (int64_t k = 0; k < npairs && bits == 2; ++k) {
			const uint8_t *a = seq_blob + off1[k], *b = seq_blob + off2[k];
			for (int x = 0; x < len1[k]; ++x) if (code2(a[x]) < 0) { bits = 8; break; }
			for (int x = 0; x < len2[k] && bits == 2; ++x) if (code2(b[x]) < 0) { bits = 8; break; }
		}
	}
	if (bits != 2 && bits != 8) return fail(nullptr, AT_ERR_ARG, "at_pack_batch: bits must be 0, 2 or 8");
	if (bits == 2) {   /* a caller that forces 2-bit words must hand over pure ACGT: there is no code for anything else */
		for (int64_t k = 0; k < npairs; ++k) {
			const uint8_t *a = seq_blob + off1[k], *b = seq_blob + off2[k];
			for (int x = 0; x < len1[k]; ++x) if (code2(a[x]) < 0) return fail(nullptr, AT_ERR_ARG, "at_pack_batch: pair %lld: byte 0x%02x cannot be packed in 2 bits", (long long)k, a[x]);
			for (int x = 0; x < len2[k]; ++x) if (code2(b[x]) < 0) return fail(nullptr, AT_ERR_ARG, "at_pack_batch: pair %lld: byte 0x%02x cannot be packed in 2 bits", (long long)k, b[x]);
		}
	}
	if (bits_out) *bits_out = bits;
	if (!words_out) return AT_OK;   /* query only */
	const int bpw = 32 / bits;
	int64_t w = 0;
	for (int64_t k = 0; k < npairs; ++k) {
		woff1_out[k] = w;
		pack_one(seq_blob + off1[k], len1[k], bits, words_out + w);
		w += (len1[k] + bpw - 1) / bpw + 1;
		woff2_out[k] = w;
		pack_one(seq_blob + off2[k], len2[k], bits, words_out + w);
		w += (len2[k] + bpw - 1) / bpw + 1;
	}
	for (int x = 0; x < 4; ++x) words_out[w + x] = 0;
	return AT_OK;
}

extern "C" int at_render(const uint8_t *ops, int32_t nops, const uint8_t *s1, int32_t end_i,
                         const uint8_t *s2, int32_t end_j, char *r1, char *r2)
{
	if (nops < 0 || !r1 || !r2 || (nops > 0 && (!ops || !s1 || !s2))) return AT_ERR_ARG;
	int i = end_i, j = end_j;
	for (int k = 0; k < nops; ++k) {
		const int pos = nops - 1 - k;
		switch (ops[k]) {
		case AT_OP_MID: if (i <= 0 || j <= 0) return AT_ERR_ARG; r1[pos] = (char)s1[--i]; r2[pos] = (char)s2[--j]; break;
		case AT_OP_LOW: if (i <= 0) return AT_ERR_ARG; r1[pos] = (char)s1[--i]; r2[pos] = '-'; break;
		case AT_OP_UPP:
		case AT_OP_JUMP: if (j <= 0) return AT_ERR_ARG; r1[pos] = '-'; r2[pos] = (char)s2[--j]; break;
		default: return AT_ERR_ARG;
		}
	}
	r1[nops] = 0; r2[nops] = 0;
	return AT_OK;
}

/* ----------------------------------------------------------------- dispatch */

static long long env_ll(const char *name, long long dflt)
{
	const char *v = getenv(name);
	return v && *v ? atoll(v) : dflt;
}

struct Layout {
	int off_bound, off_ptr, off_sm, nsm, k, ptr_lanes;
	long long words;
};

/* rows per lane: enough to hold max_l1 in one strip of 64 lanes, at most 4 */
static int rows_per_lane(int max_l1, bool deep = false)
{
	const long long forced = getenv("AT_ROWS_PER_LANE") ? atoll(getenv("AT_ROWS_PER_LANE")) : 0;
	if (forced >= 1 && forced <= 4) return (int)forced;
	const int need = std::max(1, (max_l1 + 63) / 64);
	/* single-state kernels without a pointer matrix (overlap scores, edit) hold one value per row: 8 or 16 rows per
	 * lane spread the per-step work (shuffles, end-cell scan, loop) over more cells and fit 1 kbp in one strip */
	if (deep && need > 4 && !(forced == 8 || forced == 16)) return need <= 8 ? 8 : 16;
	if (deep && (forced == 8 || forced == 16)) return (int)forced;
	return std::min(4, need);
}

static Layout layout_for(int kmode, int bits, bool tb, int max_l1, int max_l2)
{
	const int bpw = 32 / bits;
	const int tbk = (max_l2 + 63 + at::kBlk - 1) / at::kBlk;
	const int rpb = kmode == at::K_FITJ ? 2 : 1;
	Layout L;
	L.k = rows_per_lane(max_l1, (kmode == at::K_OVERLAP && !tb) || kmode == at::K_EDIT);
	L.ptr_lanes = std::max(1, std::min(64, (max_l1 + L.k - 1) / L.k));
	const long long nstrips = (max_l1 + 64 * L.k - 1) / (64 * L.k);
	(void)bpw;
	long long nref = (at::kPad + (long long)tbk * at::kBlk) / 4 + 4;   /* s2 staged one byte per base */
	nref = (nref + 1) & ~1LL;
	const long long nbound = 2LL * (max_l2 + 2);
	const long long nptr = (tb && kmode != at::K_EDIT) ? nstrips * tbk * rpb * L.k * L.ptr_lanes + 64 : 0;
	const long long nsm = kmode == at::K_FITJ ? (((long long)max_l2 + 64 + 64 + 128 + 31) / 32 + 2 + 1) & ~1LL : 0;
	L.off_bound = (int)nref;
	L.off_sm = (int)(nref + nbound);
	L.nsm = (int)nsm;
	L.off_ptr = (int)(nref + nbound + nsm);
	L.words = nref + nbound + nsm + nptr;
	return L;
}


/* The 2-bit kernels read scores from a signed-byte LUT: scaled match/mismatch (minus the gap for overlap) must fit. */
static bool scores_fit_byte(const at_handle *h, int mode)
{
	long long a, b, c = 0, d = 0;
	if (mode == AT_MODE_EDIT) { a = 0; b = h->u; c = (long long)h->u - 2; }   /* (the sweep's table holds cost - 2: at_sweep.hip.h, RAMP) */
	else if (mode == AT_MODE_OVERLAP) {
		a = 16LL * (h->m - h->o); b = 16LL * (h->u - h->o);                     /* with pointers: s - o, scaled */
		c = (long long)h->m - 2LL * h->o; d = (long long)h->u - 2LL * h->o;     /* scores only: s - 2 o, unscaled */
	} else { a = 16LL * h->m; b = 16LL * h->u; }
	return std::llabs(a) <= 127 && std::llabs(b) <= 127 && std::llabs(c) <= 127 && std::llabs(d) <= 127;
}

/* ---- packed int16 path (at_sweep16.hip.h): two same-shape pairs per wave ---- */
struct Layout16 {
	int off_refb, off_bound, off_ptr, off_sm, nsm, g, k, ptr_lanes;
	long long words;
};

/* Group width: reads of up to 152 bases run as 8 groups of 8 lanes x K rows (16 alignments per wave: 94 % of the
 * lane-steps inside a 150 x 150 matrix, the per-step overhead spread over 19 rows; reads of up to 76 bases: 16 groups of 4
 * lanes, 32 alignments per wave), up to 208 bases as 4 groups of 16
 * lanes (8 alignments per wave, 85 %; one group of 64 lanes: 59 %) and so do 209..304 bases with 16 / 19 rows per lane,
 * 305..608 bases as 2 groups of 32 lanes (10 .. 19 rows per lane), everything
 * else as one group of 64 lanes.  AT_GROUP = 8 / 16 / 32 / 64 caps the choice (A/B runs); ragged frames name their group width themselves. */
/* The 8-lane groups x 19 rows (reads of 129 .. 152 bases): which batches take the two-pass kernels with the walk kernel by default --
 * fit (with or without -s) against a second sequence at least one and a half times as long: the sweep is long against the walkers' chain
 * of rounds (150-base reads, launches in flight / one at a time against the one-pass kernels, -s / without: l2 = 160 +12 % / -2 %, +7 % /
 * -3 %; l2 = 225 +13 % / +4 %, +8 % / +7 %; l2 = 300 +12 % / +10 %, +12 % / +11 %; l2 = 400 +12 % / +14 %, +16 % / +17 %; r05n)
 * (same box, launches in flight / one at a time: C4 150 x 500 -s 2 210 -> 2 380 / 1 960 -> 2 180 GCUPS, the same without -s 3 150 -> 3 750 /
 * 2 590 -> 3 350).  Global and local 150 x 150 gain with launches in flight (2 470 -> 2 700, 3 050 -> 3 180) and lose alone (2 380 ->
 * 2 230, 2 800 -> 2 490): AT_TWO_PASS=2 AT_TP_SPLIT=1 asks for them. */
static bool tp_split_narrow_default(int kmode, int l1, int l2) { return (kmode == at::K_FITJ || kmode == at::K_FIT) && 2LL * l2 >= 3LL * l1; }
static Layout16 layout16_for(bool tb, bool hasj, int l1, int l2, int ts, int force_g = 0, bool overlap = false, int kmode = -1, int force_k = 0, int two_pass = 0)   /* two_pass: 1 = the rounds inside the sweep's kernel (its staging area and walkers' tiles in LDS), 2 = a walk kernel; force_g: 8 / 16 = ragged frames on that group width; 64 = the 64-lane items behind a batch of narrow-group items, force_k rows per lane */
{
	Layout16 L;
	const long long g_forced = env_ll("AT_GROUP", 0);
	L.g = 64;
	L.k = rows_per_lane(l1);
	/* scores only, scores x4: 16 rows per lane (strips of 1 024 rows) once that is fewer lane-steps than strips of 256 --
	 * per step a lane does 16 x 11 + 25 instructions instead of 4 x 11 + 25 */
	if (!tb && (ts == 2 || two_pass) && !getenv("AT_ROWS_PER_LANE") && ((l1 + 1023) / 1024) * (16 * 11 + 25) < ((l1 + 255) / 256) * (4 * 11 + 25)) L.k = 16;
	/* overlap (one state: few registers even with pointers): 4 or 16 rows per lane, whichever needs fewer instructions */
	if (overlap) L.k = env_ll("AT_ROWS_PER_LANE", 0) == 4 ? 4 : ((l1 + 1023) / 1024) * (16 * 9 + 45) < ((l1 + 255) / 256) * (4 * 9 + 30) ? 16 : 4;
	if (force_g == 64) {
		/* the sliver of a batch behind its whole rounds (align_device), ragged overlap: one group of 64 lanes, whatever the read length */
		if (force_k) L.k = force_k;
	} else if (force_g == 32) {
		/* ragged frames of reads of 305 .. 608 bases, or (force_k) the sliver behind a batch of narrow-group items: one strip on two
		 * groups of 32 lanes */
		L.g = 32;
		L.k = force_k ? force_k : l1 <= 320 ? 10 : l1 <= 384 ? 12 : l1 <= 416 ? 13 : l1 <= 512 ? 16 : 19;
	} else if (!force_g && (g_forced == 0 || g_forced == 4) && ts == 4 && l1 <= 76) {
		/* reads of up to 76 bases: sixteen groups of 4 lanes x 9 / 10 / 13 / 16 / 19 rows, 32 alignments per wave */
		L.g = 4;
		L.k = l1 <= 36 ? 9 : l1 <= 40 ? 10 : l1 <= 52 ? 13 : l1 <= 64 ? 16 : 19;
	} else if (force_g != 16 && (force_g == 8 || g_forced == 0 || g_forced == 8) && ts == 4 && l1 <= 152) {
		L.g = 8;
		L.k = l1 <= 40 ? 5 : l1 <= 48 ? 6 : l1 <= 56 ? 7 : l1 <= 64 ? 8 : l1 <= 80 ? 10 : l1 <= 104 ? 13 : l1 <= 128 ? 16 : 19;
	} else if ((force_g == 16 || g_forced != 64) && ts == 4 && l1 <= 208) {
		L.g = 16;
		L.k = l1 <= 64 ? 4 : l1 <= 80 ? 5 : l1 <= 96 ? 6 : l1 <= 112 ? 7 : (l1 <= 160 ? 10 : 13);
	} else if ((force_g == 16 || (!force_g && g_forced != 64 && g_forced != 32)) && ts == 4 && l1 > 208 && l1 <= 304) {
		/* 250- and 300-base reads: still four groups of 16 lanes, 16 or 19 rows per lane (8 alignments per wave; AT_GROUP=32: the
		 * two 32-lane groups below) */
		L.g = 16;
		L.k = l1 <= 256 ? 16 : 19;
	} else if (g_forced != 64 && ts == 4 && l1 > 208 &&
	           l1 <= (force_g || g_forced == 32 ? 416 : kmode == at::K_LOCAL ? 608 : kmode == at::K_GLOBAL ? 512 : 416)) {
		/* 250- and 300-base reads: two groups of 32 lanes (4 alignments per wave); one group of 64 lanes would carry 2 and
		 * cut 300 rows into a strip of 256 and one of 44 */
		L.g = 32;
		/* 305 .. 608 bases: still one strip -- 10, 12, 13, 16 or 19 rows per lane (AT_GROUP=32: the round-1 classes only, 417+ on the
		 * 64-lane groups).  16 rows only for local and global, 19 only for local: the others spill there (268 .. 612 bytes of
		 * scratch per lane) and lose to the strips of the 64-lane group (global 560 / 608 bases: -4 %) */
		L.k = l1 <= 224 ? 7 : l1 <= 256 ? 8 : l1 <= 320 ? 10 : l1 <= 384 && g_forced != 32 ? 12 : l1 <= 416 ? 13 : l1 <= 512 ? 16 : 19;
	}
	const int ng = 64 / L.g;
	const int blk = L.g <= 16 ? 4 : 8;        /* BLK of at_sweep16 */
	const int tbk = (l2 + L.g - 1 + blk - 1) / blk;
	L.ptr_lanes = L.g == 64 ? std::max(1, std::min(64, (l1 + L.k - 1) / L.k)) : 64;
	L.ptr_lanes = (L.ptr_lanes + 3) & ~3;     /* the HBM slot stores 16 bytes per lane: keep every group of rows aligned */
	const long long nstrips = (l1 + L.g * L.k - 1) / (L.g * L.k);
	long long nref = (at::kPad + (long long)tbk * blk) / 4 + 4;
	nref = (nref + 1) & ~1LL;
	/* (two-pass tracebacks: the boundary row's words double as the replay's staging area and the walkers' tile cache) */
	/* (... the replay's rows above, CK + 1 entries of two words per lane; then every walker's copy of the block it walks) */
	const long long walk_words = ((L.k + 3) / 4 + (hasj ? ((L.k + 3) / 4 + 3) / 4 : 0)) * (long long)at::ck_steps(L.g);
	const long long nbound = std::max<long long>(2LL * (l2 + 2), two_pass == 1 ? std::max<long long>(at::ck_stage_words(at::ck_steps(L.g)), 2 * ng * walk_words * (L.g == 64 ? 4 : 1)) : 0);
	/* steps per pointer word (and alignment): 4-bit cells, 4; the jump state with scores x4 keeps byte cells, 2 -- with scores x16 it has
	 * 4-bit cells plus a bit plane of one word per 4 rows x 4 steps behind the cells (at_sweep16.hip.h: JPL); packed overlap: 2-bit cells, 8 */
	const int spw = overlap ? 16 / AT_OVL_BITS : (hasj && !(ts == 4 && AT_JPLANE)) ? 2 : 4;
	const long long njpl = tb && hasj && ts == 4 && AT_JPLANE ? nstrips * tbk * (blk / 4) * ((L.k + 3) / 4) * L.ptr_lanes : 0;
	const long long nptr = tb ? nstrips * tbk * (blk / spw) * L.k * L.ptr_lanes + njpl + 64 : 0;
	const long long nsm = hasj ? (((long long)l2 + 64 + 64 + 128 + 31) / 32 + 2 + 1) & ~1LL : 0;
	L.off_refb = (int)nref;
	L.off_bound = (int)(2 * nref * ng);
	L.off_sm = (int)(2 * nref * ng + nbound);
	L.nsm = (int)nsm;
	L.off_ptr = (int)((2 * nref * ng + nbound + nsm + 3) & ~3LL);   /* 16-byte aligned: wide pointer stores */
	L.words = L.off_ptr + nptr;
	return L;
}

/* Scores of every real (non -inf) cell lie in [-lo, hi]; the packed kernel needs
 * 16*(lo + hi) plus slack below 2^15 so that the -32768 sentinel, even after the
 * hi*16 it can gain along a diagonal of matches, stays below every real value. */
static bool packed_ok(const at_handle *h, int mode, int bits, int l1, int l2, int ts, int *thresh16)
{
	const long long scale = 1LL << ts;
	if (getenv("AT_NO_PACKED") && atoi(getenv("AT_NO_PACKED"))) return false;
	if ((bits != 2 && bits != 8) || l1 < 1 || l2 < 1) return false;   /* 2-bit codes: score LUT; bytes: compare */
	if (mode == AT_MODE_OVERLAP) {
		/* one state, linear gap (alignment.h:926-964; -e unused): scores x4 with 2-bit tags only.  A cell's value is at least
		 * that of the path along its row from column 0 (where M = 0): -|o| * j; at most m * min(l1, l2) */
		if (ts != 2 || h->m < 0 || h->u > 0 || h->o > 0) return false;
		const long long lo = std::llabs((long long)h->o) * std::max(l1, l2) + std::llabs((long long)h->u) + 16;
		const long long hi = (long long)h->m * std::min(l1, l2), slack = std::llabs((long long)h->o) + std::llabs((long long)h->u) + 3;
		if (scale * (lo + hi + slack) >= 32768) return false;
		*thresh16 = (int)(-32768 + scale * (hi + slack));
		return true;
	}
	if (!(mode == AT_MODE_GLOBAL || mode == AT_MODE_LOCAL || mode == AT_MODE_FIT)) return false;
	if (h->m < 0 || h->u > 0 || h->o > 0 || h->e > 0) return false;
	const bool hasj = mode == AT_MODE_FIT && h->use_jump;
	if (hasj && (h->j > 0 || std::llabs((long long)h->j - h->o) * scale > 32000)) return false;
	const long long A = std::max<long long>(std::max(std::llabs((long long)h->e), std::llabs((long long)h->u)), h->m);
	/* global / fit: a cell's value is a max over paths, hence at least the value of one of them -- min(i, j) diagonal
	 * steps, each >= u, and one gap of |i - j|, >= o + e * |i - j|; the L and U states lie at most one more opening below */
	(void)A;
	long long lo = 3 * std::llabs((long long)h->o) + (hasj ? std::llabs((long long)h->j) : 0) +
	               std::llabs((long long)h->u) * std::min(l1, l2) + std::llabs((long long)h->e) * std::max(l1, l2) + 16;
	/* local: M >= 0 everywhere (the 0 candidate, alignment.h:826), so L and U, each the max of something and M + o, never
	 * drop below o, and no cell or intermediate below o + u + e: the matrix has no -inf and no downward drift */
	if (mode == AT_MODE_LOCAL) lo = std::llabs((long long)h->o) + std::llabs((long long)h->u) + std::llabs((long long)h->e) + 16;
	const long long hi = (long long)h->m * std::min(l1, l2);
	const long long slack = std::llabs((long long)h->o) + std::llabs((long long)h->e) + 3;
	if (scale * (lo + hi + slack) >= 32768) return false;
	*thresh16 = (int)(-32768 + scale * (hi + slack));
	return true;
}

/* Two-pass tracebacks (at_sweep16.hip.h, CK kernels): the regions of a wave's global slot -- border row, row checkpoints, column
 * checkpoints, the replayed blocks' pointer words and jump plane -- for the scores-only layout L of the sweep */
struct TpLayout {
	int off_brow, off_rck, off_cck, off_rptr, off_rjpl;
	long long words;
	long long brow_words;
};
/* split: pass 2 is a kernel of its own -- the regions of a work item (row and column checkpoints); the border row is the launch's */
static TpLayout tp_layout(const Layout16 &L, int kmode, int l2, bool split = false)
{
	const int blk = L.g <= 16 ? 4 : 8, cb = at::ck_steps(L.g);
	const long long T = (long long)((l2 + L.g - 1 + blk - 1) / blk) * blk;   /* steps of a sweep */
	const int es = 2, nq = ((kmode == at::K_FITJ ? 3 : 2) * L.k + 3) / 4, kg = (L.k + 3) / 4;
	auto up4 = [](long long v) { return (v + 3) & ~3LL; };
	TpLayout t;
	long long w = 0;
	t.brow_words = (T + cb + 2) * es;
	t.off_brow = (int)w; if (!split) w = up4(w + (T + cb + 2) * es);
	t.off_rck = (int)w; w = up4(w + 64 * es + (T / cb + 3) * 64 * cb * es);   /* (ck_rck_word: the border entries, then tiles of steps) */
	t.off_cck = (int)w; w = up4(w + (T / cb + 3) * ((nq + AT_CK_COL_QUAD - 1) / AT_CK_COL_QUAD * AT_CK_COL_QUAD) * 256);   /* (ck_cck_word: chunks in groups of AT_CK_COL_QUAD) */
	t.off_rptr = (int)w; if (!split) w = up4(w + (cb / 4) * ((L.k + 3) / 4) * 256);   /* [4 steps][4 rows][lane][row in group] */
	t.off_rjpl = (int)w; if (!split) w = up4(w + (cb / 4) * ((kg + 3) / 4) * 256);
	t.words = split ? (w + 63) & ~63LL : w;
	return t;
}

static int choose_store(long long words_fixed, long long words_ptr, bool prefer_hbm_pointers);
/* is there a packed instantiation for the storage class plan_launch will choose for this layout? */
static bool packed16_kernel_exists(int kmode, const Layout16 &P, bool tb, int ts, int bits, int rag, const Layout16 *PT = nullptr)
{
	/* (the launch plans with the larger of the main items' layout and the sliver items': the same maximum here) */
	const long long fixed = PT ? std::max(P.off_ptr, PT->off_ptr) : P.off_ptr;
	const long long ptrw = PT ? std::max(P.words - P.off_ptr, PT->words - PT->off_ptr) : P.words - P.off_ptr;
	const int st = choose_store(fixed, ptrw, P.g < 64 || rag);   /* (ragged frames exist with the pointers in the global slots only) */
	if (P.g != 64 && st == 2) return false;
	return (rag ? at_pick16_rag(kmode, P.g, P.k, st, tb, bits) : at_pick16(kmode, P.g, P.k, ts, st, tb, bits)) != nullptr;
}

static int grow(at_handle *h, void **p, size_t *have, size_t need)
{
	if (need <= *have) return AT_OK;
	if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
	need = need + need / 8 + 4096;
	hipError_t e = hipMalloc(p, need);
	if (e != hipSuccess) return fail(h, AT_ERR_NOMEM, "hipMalloc(%zu): %s", need, hipGetErrorString(e));
	*have = need;
	return AT_OK;
}

/* page-locked host memory, grow-only */
static int grow_pinned(at_handle *h, void **p, size_t *have, size_t need)
{
	if (need <= *have) return AT_OK;
	if (*p) { (void)hipHostFree(*p); *p = nullptr; *have = 0; }
	need = need + need / 4 + 4096;
	hipError_t e = hipHostMalloc(p, need, hipHostMallocDefault);
	if (e != hipSuccess) return fail(h, AT_ERR_NOMEM, "hipHostMalloc(%zu): %s", need, hipGetErrorString(e));
	*have = need;
	return AT_OK;
}

static int ensure_sitemask(at_handle *h, int max_l2, hipStream_t stream)
{
	if (!h->sitemask_dirty && h->sitemask_for_l2 >= max_l2) return AT_OK;
	const size_t nbits = (size_t)max_l2 + 64 + 64 + 128;
	const size_t nw = (nbits + 31) / 32 + 2;
	std::vector<uint32_t> m(nw, 0xffffffffu);
	/* column j (1-based) may open the jump state iff (j-1) is NOT a listed site
	 * -- the reference's inverted test, alignment.h:659 / SURVEY.md 0.4 */
	for (int sidx : h->sites) {
		const long long j = (long long)sidx + 1;
		if (j < 0 || (size_t)(j + 64) >= nbits) continue;
		m[(size_t)(j + 64) >> 5] &= ~(1u << ((j + 64) & 31));
	}
	void *p = h->d_sitemask; size_t have = h->sitemask_words * 4;
	int rc = grow(h, &p, &have, nw * 4);
	h->d_sitemask = (uint32_t *)p; h->sitemask_words = have / 4;
	if (rc) return rc;
	HIP_TRY(h, hipMemcpyAsync(h->d_sitemask, m.data(), nw * 4, hipMemcpyHostToDevice, stream));
	HIP_TRY(h, hipStreamSynchronize(stream));   /* m is a temporary */
	h->sitemask_dirty = false;
	h->sitemask_for_l2 = max_l2;
	return AT_OK;
}



/* Storage class + grid for one launch.
 *   store 0  s2 window, boundary row and pointer matrix in LDS
 *   store 1  s2 + boundary in LDS, pointer matrix in a per-wave global slot (HBM/L2)
 *   store 2  everything in the global slot (very long s2)
 * Work is handed out through an atomic counter, so the grid only has to cover the waves that
 * can be resident; over-estimating it is harmless. */
struct Plan {
	int store, off_ptr;
	long long grid, slot_words;
	size_t dyn_lds;
	uint32_t *ws;
};

/* storage class for (fixed, pointer) words per work item; no side effects (also used to ask whether a kernel exists) */
static int choose_store(long long words_fixed, long long words_ptr, bool prefer_hbm_pointers)
{
	/* all-LDS only while at least 8 waves (2 per SIMD) still fit a CU: measured, occupancy beyond 1 wave/SIMD is
	 * worth +30..50 % on this issue-bound kernel (profiles/r01) */
	const long long limit_all = env_ll("AT_SMALL_LDS_LIMIT", 20 * 1024);
	const long long limit_fixed = env_ll("AT_MEDIUM_LDS_LIMIT", 64 * 1024);
	const long long forced = env_ll("AT_STORE", -1);
	int store;
	if (forced >= 0 && forced <= 2) store = (int)forced;
	else if ((words_fixed + words_ptr) * 4 <= limit_all) store = 0;
	else if (words_fixed * 4 <= limit_fixed) store = 1;
	else store = 2;
	/* several alignments per wave (16- and 32-lane groups): the pointer matrices in LDS would cap the CU at a few waves
	 * even when they fit (49-base reads: 1 151 GCUPS all-LDS, 1 357 with the slots in HBM) */
	if (prefer_hbm_pointers && forced < 0 && store == 0 && words_ptr > 0) store = 1;
	if (words_ptr == 0 && store == 1) store = 0;
	if (store < 2 && words_fixed * 4 > 150 * 1024) store = 2;
	if (store == 0 && (words_fixed + words_ptr) * 4 > 150 * 1024) store = 1;
	return store;
}

static int plan_launch(at_handle *h, const char *tag, int k, long long nwork, long long words_fixed, long long words_ptr,
                       Plan *pl, hipStream_t stream, const std::function<const void *(int)> &kernel_for_store,
                       bool prefer_hbm_pointers = false)
{
	const int store = choose_store(words_fixed, words_ptr, prefer_hbm_pointers);
	const long long lds_words = store == 0 ? words_fixed + words_ptr : (store == 1 ? words_fixed : 0);
	const long long slot_words = store == 0 ? 0 : (((store == 1 ? words_ptr : words_fixed + words_ptr) + 63) & ~63LL);
	long long per_cu = 16;
	if (lds_words > 0) per_cu = std::min<long long>(per_cu, (long long)(h->lds_per_cu - 512) / std::max<long long>(lds_words * 4, 256));
	per_cu = std::max(1LL, std::min(per_cu, env_ll("AT_WAVES_PER_CU", 16)));
	{   /* what the register file really admits */
		int occ = 0;
		const void *fn = kernel_for_store ? kernel_for_store(store) : nullptr;
		if (fn && hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 64, (size_t)lds_words * 4) == hipSuccess && occ > 0)
			per_cu = std::min<long long>(per_cu, occ);
	}
	long long grid = std::max(1LL, std::min<long long>(nwork, per_cu * h->ncu));
	/* (Tried and dropped: shrinking the grid so that every wave gets the same number of items -- fewer resident
	 * waves cost 5-13 % more than the partial last round they avoid.) */
	pl->store = store;
	pl->off_ptr = store == 1 ? 0 : (int)words_fixed;
	pl->dyn_lds = (size_t)lds_words * 4;
	pl->slot_words = slot_words;
	pl->ws = nullptr;
	if (slot_words > 0) {
		const long long cap = env_ll("AT_WS_CAP_MB", 16384) << 20;
		if (slot_words * 4 > cap) return fail(h, AT_ERR_NOMEM, "one pair needs %lld workspace bytes (cap %lld)", slot_words * 4, cap);
		grid = std::max(1LL, std::min(grid, cap / (slot_words * 4)));
		void *p = h->d_ws; size_t have = h->ws_bytes;
		int rc = grow(h, &p, &have, (size_t)(grid * slot_words * 4));
		h->d_ws = (uint32_t *)p; h->ws_bytes = have;
		if (rc) return rc;
		pl->ws = h->d_ws;
	}
	pl->grid = grid;
	if (!h->d_queue) { HIP_TRY(h, hipMalloc((void **)&h->d_queue, 128)); HIP_TRY(h, hipMemset(h->d_queue, 0, 128)); }
	HIP_TRY(h, hipMemsetAsync(h->d_queue, 0, 8, stream));
	static const char *names[3] = {"lds", "lds+hbm-pointers", "hbm"};
	snprintf(h->cfg, sizeof h->cfg, "%s store=%s rows/lane=%d lds=%zuB slot=%lldB waves/cu<=%lld grid=%lld", tag, names[store], k,
	         pl->dyn_lds, slot_words * 4, per_cu, grid);
	return AT_OK;
}

static int align_device(at_handle *h, int mode, int64_t npairs,
                        const uint32_t *d_seq, int bits,
                        const int64_t *d_woff1, const int32_t *d_len1,
                        const int64_t *d_woff2, const int32_t *d_len2,
                        int32_t max_len1, int32_t max_len2, int uniform_shape, int want_traceback,
                        int32_t *d_score, int32_t *d_end_i, int32_t *d_end_j, int32_t *d_state,
                        uint8_t *d_ops, const int64_t *d_ops_off, int32_t *d_nops, void *stream_,
                        int64_t ap_n, int64_t ap_first, const int *d_order = nullptr, int rag = 0,
                        const int *only_if = nullptr, int only_val = 0);

extern "C" int at_align_batch_device(at_handle *h, int mode, int64_t npairs,
                                     const uint32_t *d_seq, int bits,
                                     const int64_t *d_woff1, const int32_t *d_len1,
                                     const int64_t *d_woff2, const int32_t *d_len2,
                                     int32_t max_len1, int32_t max_len2, int uniform_shape, int want_traceback,
                                     int32_t *d_score, int32_t *d_end_i, int32_t *d_end_j, int32_t *d_state,
                                     uint8_t *d_ops, const int64_t *d_ops_off, int32_t *d_nops, void *stream_)
{
	if (!d_woff2 || !d_len2) return fail(h, AT_ERR_ARG, "NULL device pointer");
	return align_device(h, mode, npairs, d_seq, bits, d_woff1, d_len1, d_woff2, d_len2, max_len1, max_len2, uniform_shape,
	                    want_traceback, d_score, d_end_i, d_end_j, d_state, d_ops, d_ops_off, d_nops, stream_, 0, 0);
}

extern "C" int at_align_allpairs_device(at_handle *h, int mode, int64_t nreads,
                                        const uint32_t *d_seq, int bits,
                                        const int64_t *d_woff, const int32_t *d_len, int32_t max_len,
                                        int64_t first_pair, int64_t npairs, int want_traceback,
                                        int32_t *d_score, int32_t *d_end_i, int32_t *d_end_j, int32_t *d_state,
                                        uint8_t *d_ops, const int64_t *d_ops_off, int32_t *d_nops, void *stream_)
{
	if (!h) return fail(nullptr, AT_ERR_ARG, "at_align_allpairs_device: NULL handle");
	if (nreads < 2 || first_pair < 0 || npairs < 0 || first_pair + npairs > nreads * (nreads - 1) / 2)
		return fail(h, AT_ERR_ARG, "all-vs-all: pair range [%lld, +%lld) outside the %lld*(%lld-1)/2 ordered pairs",
		            (long long)first_pair, (long long)npairs, (long long)nreads, (long long)nreads);
	return align_device(h, mode, npairs, d_seq, bits, d_woff, d_len, d_woff, d_len, max_len, max_len, 0, want_traceback,
	                    d_score, d_end_i, d_end_j, d_state, d_ops, d_ops_off, d_nops, stream_, nreads, first_pair);
}

extern "C" int at_render_batch_device(at_handle *h, int64_t npairs,
                                      const uint32_t *d_seq, int bits,
                                      const int64_t *d_woff1, const int64_t *d_woff2,
                                      const int32_t *d_end_i, const int32_t *d_end_j,
                                      const uint8_t *d_ops, const int64_t *d_ops_off, const int32_t *d_nops,
                                      uint8_t *d_r1, uint8_t *d_r2, const int64_t *d_str_off, int nul_terminate,
                                      void *stream_)
{
	if (!h) return fail(nullptr, AT_ERR_ARG, "at_render_batch_device: NULL handle");
	if (npairs < 0) return fail(h, AT_ERR_ARG, "negative npairs");
	if (npairs == 0) return AT_OK;
	if (bits != 2 && bits != 8) return fail(h, AT_ERR_ARG, "bits must be 2 or 8");
	if (!d_seq || !d_woff1 || !d_woff2 || !d_end_i || !d_end_j || !d_ops || !d_ops_off || !d_nops || !d_r1 || !d_r2)
		return fail(h, AT_ERR_ARG, "NULL device pointer");
	HIP_TRY(h, hipSetDevice(h->device));
	hipStream_t s = (hipStream_t)stream_;
	at::RenderArgs ra;
	ra.npairs = npairs; ra.seq = d_seq; ra.woff1 = (const long long *)d_woff1; ra.woff2 = (const long long *)d_woff2;
	ra.end_i = d_end_i; ra.end_j = d_end_j; ra.ops = d_ops; ra.ops_off = (const long long *)d_ops_off; ra.nops = d_nops;
	ra.r1 = d_r1; ra.r2 = d_r2; ra.str_off = (const long long *)d_str_off; ra.nul = nul_terminate ? 1 : 0; ra.bad = h->d_rflag;
	/* four pairs per wavefront, 16 lanes each (AT_RENDER_GROUP = 8 / 32 / 64: A/B runs; 64 = one pair per wavefront, the round-1 form) */
	/* alignments of long pairs are hundreds of ops long: one pair per wavefront again (C3: 2 833 against 2 871 GCUPS with four) */
	const long long rw = env_ll("AT_RENDER_GROUP", h->last_span >= 1024 ? 64 : 16);
	const int w = rw == 64 ? 64 : rw == 32 ? 32 : rw == 8 ? 8 : 16;
	const int per_block = 4 * (64 / w);
	const unsigned grid = (unsigned)std::min<int64_t>((npairs + per_block - 1) / per_block, 16LL * h->ncu);
#define AT_RENDER(B, W) hipLaunchKernelGGL((at::at_render_k<B, W>), dim3(grid), dim3(256), 0, s, ra)
	if (bits == 2) { if (w == 64) AT_RENDER(2, 64); else if (w == 32) AT_RENDER(2, 32); else if (w == 8) AT_RENDER(2, 8); else AT_RENDER(2, 16); }
	else { if (w == 64) AT_RENDER(8, 64); else if (w == 32) AT_RENDER(8, 32); else if (w == 8) AT_RENDER(8, 8); else AT_RENDER(8, 16); }
#undef AT_RENDER
	HIP_TRY(h, hipGetLastError());
	return AT_OK;
}

/* exclusive prefix sums of max(nops, 0) into d_off[0 .. npairs] (d_off[npairs] = total) */
static int scan_nops_device(at_handle *h, int64_t npairs, const int32_t *d_nops, int64_t *d_off, hipStream_t s)
{
	const unsigned ntiles = (unsigned)((npairs + at::SCAN_TILE - 1) / at::SCAN_TILE);
	int rc = grow(h, &h->d_scan, &h->scan_bytes, (size_t)ntiles * 8);
	if (rc) return rc;
	hipLaunchKernelGGL(at::at_scan_tiles, dim3(ntiles), dim3(256), 0, s, d_nops, (long long)npairs, (long long *)h->d_scan);
	hipLaunchKernelGGL(at::at_scan_nops, dim3(ntiles), dim3(256), 0, s, d_nops, (long long)npairs, (const long long *)h->d_scan,
	                   (long long *)d_off);
	HIP_TRY(h, hipGetLastError());
	return AT_OK;
}

extern "C" int at_compact_ops_device(at_handle *h, int64_t npairs,
                                     const uint8_t *d_ops, const int64_t *d_ops_off, const int32_t *d_nops,
                                     uint8_t *d_packed, int64_t packed_cap, int64_t *d_packed_off, void *stream_)
{
	if (!h) return fail(nullptr, AT_ERR_ARG, "at_compact_ops_device: NULL handle");
	if (npairs < 0 || packed_cap < 0) return fail(h, AT_ERR_ARG, "negative size");
	if (!d_ops || !d_ops_off || !d_nops || !d_packed || !d_packed_off) return fail(h, AT_ERR_ARG, "NULL device pointer");
	HIP_TRY(h, hipSetDevice(h->device));
	hipStream_t s = (hipStream_t)stream_;
	if (npairs == 0) {
		HIP_TRY(h, hipMemsetAsync(d_packed_off, 0, 8, s));
		return AT_OK;
	}
	int rc = scan_nops_device(h, npairs, d_nops, d_packed_off, s);
	if (rc) return rc;
	{
		const unsigned grid = (unsigned)std::min<int64_t>((npairs + 15) / 16, 16LL * h->ncu);
		hipLaunchKernelGGL(at::at_compact_k, dim3(grid), dim3(256), 0, s, (long long)npairs, d_ops, (const long long *)d_ops_off,
		                   d_nops, d_packed, (const long long *)d_packed_off, (long long)packed_cap);
	}
	HIP_TRY(h, hipGetLastError());
	return AT_OK;
}

static int align_device(at_handle *h, int mode, int64_t npairs,
                        const uint32_t *d_seq, int bits,
                        const int64_t *d_woff1, const int32_t *d_len1,
                        const int64_t *d_woff2, const int32_t *d_len2,
                        int32_t max_len1, int32_t max_len2, int uniform_shape, int want_traceback,
                        int32_t *d_score, int32_t *d_end_i, int32_t *d_end_j, int32_t *d_state,
                        uint8_t *d_ops, const int64_t *d_ops_off, int32_t *d_nops, void *stream_,
                        int64_t ap_n, int64_t ap_first, const int *d_order, int rag, const int *only_if, int only_val)
{
	if (!h) return fail(nullptr, AT_ERR_ARG, "at_align_batch_device: NULL handle");
	if (mode < AT_MODE_GLOBAL || mode > AT_MODE_EDIT) return fail(h, AT_ERR_ARG, "unknown mode %d", mode);
	if (npairs < 0 || max_len1 < 0 || max_len2 < 0) return fail(h, AT_ERR_ARG, "negative size");
	if (bits != 2 && bits != 8) return fail(h, AT_ERR_ARG, "bits must be 2 or 8");
	if (npairs == 0) return AT_OK;
	if (!d_seq || !d_woff1 || !d_len1 || !d_woff2 || !d_len2 || !d_score) return fail(h, AT_ERR_ARG, "NULL device pointer");
	const bool tb = want_traceback && mode != AT_MODE_EDIT;
	if (tb && (!d_ops || !d_ops_off || !d_nops)) return fail(h, AT_ERR_ARG, "traceback wanted but ops buffers are NULL");
	hipStream_t stream = (hipStream_t)stream_;
	HIP_TRY(h, hipSetDevice(h->device));

	/* exact-int32 range: real scores stay within 2^24, sentinel at -2^26 (at_sweep.hip.h) */
	long long maxabs = 0;
	for (int v : {h->m, h->u, h->o, h->e, h->j}) maxabs = std::max<long long>(maxabs, std::llabs((long long)v));
	maxabs = std::max<long long>(maxabs, 1);
	if (maxabs * ((long long)max_len1 + max_len2 + 2) >= (1LL << 24))
		return fail(h, AT_ERR_RANGE, "scores may exceed the exact range: max|param|=%lld, l1+l2=%lld", maxabs,
		            (long long)max_len1 + max_len2);

	h->last_span = max_len1 + max_len2;
	const int kmode = mode == AT_MODE_GLOBAL ? at::K_GLOBAL : mode == AT_MODE_LOCAL ? at::K_LOCAL
	                : mode == AT_MODE_FIT ? (h->use_jump ? at::K_FITJ : at::K_FIT)
	                : mode == AT_MODE_OVERLAP ? at::K_OVERLAP : at::K_EDIT;

	/* ---- packed int16 path: uniform shape, scores provably within 16 bits ---- */
	/* scores x16 with nibble pointers when they fit (|score| < 2048), else x4 with byte pointers (|score| < 8192) */
	int thresh16 = 0, ts = 0;
	if (!uniform_shape && !rag && !only_if && ap_n == 0 && npairs >= 4096 && !d_order && kmode <= at::K_FITJ &&
	    env_ll("AT_AUTO_UNIFORM", 1) && (bits == 8 || scores_fit_byte(h, mode)) &&
	    packed_ok(h, mode, bits, max_len1, max_len2, 4, &thresh16)) {
		/* The caller did not promise one shape, but the batch may well have one (fixed-length reads against fixed
		 * windows).  The lengths live on the device and this entry is asynchronous, so the device decides: a check
		 * kernel leaves 1 in a flag if every pair is max_len1 x max_len2, and the packed and the int32 launch are both
		 * queued, each of them a no-op unless the flag names it. */
		int *flag = h->d_rflag + 8;
		HIP_TRY(h, hipMemsetAsync(flag, 0xff, 4, stream));
		const unsigned cg = (unsigned)std::min<int64_t>((npairs + 255) / 256, 4LL * h->ncu);
		hipLaunchKernelGGL(at::at_check_uniform, dim3(cg), dim3(256), 0, stream, d_len1, d_len2, (long long)npairs, max_len1, max_len2, flag);
		int rc = align_device(h, mode, npairs, d_seq, bits, d_woff1, d_len1, d_woff2, d_len2, max_len1, max_len2, 1, want_traceback,
		                      d_score, d_end_i, d_end_j, d_state, d_ops, d_ops_off, d_nops, stream_, 0, 0, nullptr, 0, flag, -1);
		if (rc) return rc;
		std::string packed_cfg = h->cfg;
		rc = align_device(h, mode, npairs, d_seq, bits, d_woff1, d_len1, d_woff2, d_len2, max_len1, max_len2, 0, want_traceback,
		                  d_score, d_end_i, d_end_j, d_state, d_ops, d_ops_off, d_nops, stream_, 0, 0, nullptr, 0, flag, 0);
		if (rc) return rc;
		snprintf(h->cfg, sizeof h->cfg, "auto: [%.140s] if every pair is %dx%d, else [%.120s]", packed_cfg.c_str(), max_len1, max_len2,
		         std::string(h->cfg).c_str());
		return AT_OK;
	}
	if ((uniform_shape || rag) && ap_n == 0) {
		if (packed_ok(h, mode, bits, max_len1, max_len2, 4, &thresh16)) ts = 4;
		else if ((!rag || kmode == at::K_OVERLAP) && packed_ok(h, mode, bits, max_len1, max_len2, 2, &thresh16)) ts = 2;
		/* overlap: the packed kernel exists with pointers only (scores alone: the int32 kernel's 2 instructions per cell win) */
		if (kmode == at::K_OVERLAP && (!tb || getenv("AT_NO_PACKED_OVERLAP"))) ts = 0;
	}
	if (rag && (!ts || (kmode > at::K_FITJ && !(kmode == at::K_OVERLAP && rag == 64)) || max_len1 > (rag == 8 ? 152 : rag == 16 ? 304 : rag == 32 ? 608 : 1024) || !d_order))
		return fail(h, AT_ERR_ARG, "ragged packed launch outside its domain");   /* the host entry checks before it asks */
	Layout16 P;
	if (ts) {
		P = layout16_for(tb, kmode == at::K_FITJ, max_len1, max_len2, ts, rag, kmode == at::K_OVERLAP, kmode,
		                 rag && kmode == at::K_OVERLAP ? (max_len1 <= 256 ? 4 : 16) : 0);   /* (ragged overlap: one strip) */
		/* A 64-lane packed wave carries 2 alignments where an int32 wave carries 1: fewer, longer work items.  A batch
		 * that cannot give every resident wave one of them stays on the int32 kernel.  (10k x 1024^2 = 1.6 rounds: 1.96
		 * packed vs 2.00 TCUPS int32 for a lone launch, 2.70 vs 2.13 with launches in flight; 61k pairs: 2.70 vs 2.19.) */
		const double min_rounds = getenv("AT_PACKED_MIN_ROUNDS") ? atof(getenv("AT_PACKED_MIN_ROUNDS")) : 1.0;
		if (P.g == 64 && (double)((npairs + 1) / 2) < min_rounds * 12.0 * h->ncu) ts = 0;
		/* the packed kernels index their slot with 24-bit multiplies: a pair whose slot would not fit takes the int32 kernel */
		if (P.words >= (1LL << 24)) { if (rag) return fail(h, AT_ERR_ARG, "ragged packed launch outside its domain (slot too large)"); ts = 0; }
		/* no packed kernel for the storage class this shape needs (the 16- and 32-lane groups have no all-HBM variant:
		 * a 150-base read against a second sequence of more than ~4 000 bases): the int32 kernel takes any length */
		Layout16 PTq = P;
		const bool tailq = !rag && !only_if && P.g <= 16 && kmode != at::K_OVERLAP && env_ll("AT_TAIL_SPLIT", 1);
		if (tailq) PTq = layout16_for(tb, kmode == at::K_FITJ, max_len1, max_len2, ts, at::AT_TAIL_G, false, kmode, at::at_tail_k(P.g, P.k));
		if (ts && !packed16_kernel_exists(kmode, P, tb, ts, bits, rag, tailq ? &PTq : nullptr)) {
			if (rag) return fail(h, AT_ERR_ARG, "ragged packed launch outside its domain (frame too long for LDS)");
			ts = 0;
		}
	}
	/* two-pass tracebacks: uniform batches whose shape has a CK kernel sweep without pointers and rebuild them where the walks go.
	 * By default where that wins -- the 64-lane groups with 16 rows per lane (reads of 609 .. 1 024 bases: the one-pass kernels hold 4 rows
	 * per lane there and need four strips); AT_TWO_PASS=2: wherever a CK kernel exists (the 8-lane groups x 19 rows lose: C2 -7 %, C4
	 * -28 %, profiles/r04/two_pass_ab.txt), AT_TWO_PASS=0: never (A/B runs) */
	bool two_pass = false;
	int tp_split = 0;   /* pass 2: 0 the rounds inside the sweep's kernel, 1 a walk kernel behind it */
	const long long tp_mode = env_ll("AT_TWO_PASS", 1);
	if (ts && tb && !rag && kmode <= at::K_FITJ && tp_mode) {
		Layout16 P2 = layout16_for(false, kmode == at::K_FITJ, max_len1, max_len2, ts, 0, false, kmode, 0, 1);
		/* pass 2 as a kernel of its own (at_walk16.hip.h): the sweep leaves its checkpoints per work item, walkers replay the blocks their
		 * walks cross.  By default on the 64-lane groups, where teams of walker lanes cut the chain of rounds (C3 2 940 -> 4 420 GCUPS with
		 * launches in flight, 2 150 -> 3 310 one at a time, against the one-pass kernels), and on the 8-lane groups for fit against a long
		 * second sequence (tp_split_narrow_default); AT_TP_SPLIT=1: wherever a walk kernel exists, 0: nowhere (the rounds inside the sweep's kernel) */
		const bool narrow_default = P2.g == 8 && tp_split_narrow_default(kmode, max_len1, max_len2);
		const bool split_default = P2.g == 64 || narrow_default;
		const int split_req = env_ll("AT_TP_SPLIT", split_default ? 1 : 0) && !h->ck_alloc_failed && at_pick_walk16(kmode, P2.g, P2.k, ts, bits) ? 1 : 0;
		if ((P2.g == 64 || tp_mode >= 2 || (split_req && narrow_default)) && (long long)P2.g * P2.k >= max_len1 && at_pick16_tp(kmode, P2.g, P2.k, ts, bits) &&
		    choose_store(P2.off_ptr, 1, true) == 1) {
			two_pass = true;
			tp_split = split_req;
			if (tp_split) P2 = layout16_for(false, kmode == at::K_FITJ, max_len1, max_len2, ts, 0, false, kmode, 0, 2);
			P = P2;
		}
	}
	if (ts) {
		const int per_wave = 2 * (64 / P.g);
		/* the sliver's layout (see below) */
		const bool tail_ok = !rag && !only_if && P.g <= 16 && kmode != at::K_OVERLAP && env_ll("AT_TAIL_SPLIT", 1) &&
		                     (!tp_split || P.g == 8);   /* (the walk kernels of the 8-lane groups carry their sliver's walks) */
		Layout16 PT = P;
		if (tail_ok) PT = layout16_for(tb && !two_pass, kmode == at::K_FITJ, max_len1, max_len2, ts, at::AT_TAIL_G, false, kmode, at::at_tail_k(P.g, P.k), two_pass ? (tp_split ? 2 : 1) : 0);
		TpLayout tpm{}, tpt{};
		if (two_pass) { tpm = tp_layout(P, kmode, max_len2, tp_split != 0); tpt = tp_layout(PT, kmode, max_len2, tp_split != 0); }
		/* split two-pass: the checkpoints of a whole launch live side by side -- C2 225 KB, C3 / C4 1.1 MB per work item -- so a batch is
		 * swept and walked in pieces whose checkpoints fit AT_CK_CAP_MB (8 192) */
		int64_t piece = npairs;
		int64_t brow_words = 0;
		if (tp_split) {
			const long long cap = env_ll("AT_CK_CAP_MB", 8192) << 20;
			brow_words = (tpm.brow_words + 63) & ~63LL;
			const long long tail_reserve = tail_ok ? 16LL * h->ncu * tpt.words * 4 : 0;
			const long long items = (cap - tail_reserve - brow_words * 4) / (tpm.words * 4 + 16LL * per_wave);
			if (items < 1) return fail(h, AT_ERR_NOMEM, "two-pass tracebacks: one work item's checkpoints (%lld bytes) exceed AT_CK_CAP_MB", tpm.words * 4);
			piece = std::min<int64_t>(npairs, items * per_wave);
			const long long forced = env_ll("AT_CK_PIECE_PAIRS", 0);   /* (tests: pieces of this many pairs, whatever the cap) */
			if (forced > 0) piece = std::min<int64_t>(piece, std::max<long long>(1, forced / per_wave) * per_wave);
		}
		auto walk_lane_words = [&](const Layout16 &L) -> long long {   /* a walker lane's pointer words of one block (at_walk16.hip.h) */
			const int cb = at::ck_steps(L.g);
			return L.k * (cb / 4) + (kmode == at::K_FITJ ? (L.k + 3) / 4 * (cb / 4) : 0);
		};
		auto walk_lds = [&](const Layout16 &L, bool ptr_in_lds) -> size_t {
			const int cb = at::ck_steps(L.g), nsm = kmode == at::K_FITJ ? L.nsm : 0;
			return (size_t)((nsm + 1) / 2 * 2 + (L.k > 16 ? 0 : (cb + 1) * 128) + (ptr_in_lds ? 64 * walk_lane_words(L) : 0)) * 4;   /* (walk16_lds_words) */
		};
		std::string cfg_first;
		for (int64_t first = 0; first < npairs; first += piece) {
		const int64_t np = std::min<int64_t>(piece, npairs - first);
		Sweep16Args b;
		memset(&b, 0, sizeof b);
		b.npairs = np; b.seq = d_seq;
		b.woff1 = (const long long *)d_woff1 + first; b.woff2 = (const long long *)d_woff2 + first;
		b.len1 = d_len1 + first; b.len2 = d_len2 + first;
		b.l1 = max_len1; b.l2 = max_len2;
		const int sc = 1 << ts;
		b.m16 = h->m * sc; b.u16 = h->u * sc; b.o16 = h->o * sc; b.e16 = h->e * sc; b.g16 = h->j * sc; b.thresh16 = thresh16;
		if (kmode == at::K_FITJ) {
			int rcs = ensure_sitemask(h, max_len2, stream);
			if (rcs) return rcs;
			b.sitemask = h->d_sitemask;
		}
		b.score = d_score + first; b.end_i = d_end_i ? d_end_i + first : nullptr; b.end_j = d_end_j ? d_end_j + first : nullptr;
		b.state = d_state ? d_state + first : nullptr;
		b.ops = d_ops; b.ops_off = d_ops_off ? (const long long *)d_ops_off + first : nullptr; b.nops = d_nops ? d_nops + first : nullptr;
		b.order = rag ? d_order : nullptr;
		b.only_if = only_if; b.only_val = only_val;
		b.off_refb = P.off_refb; b.off_bound = P.off_bound; b.ptr_lanes = P.ptr_lanes; b.off_sm = P.off_sm; b.nsm = P.nsm;
		Plan pl;
		char tag16[128];
		snprintf(tag16, sizeof tag16, "packed16 x%d bits=%d %dx%d-lane groups (%d pairs/wave)%s", 1 << ts, bits, 64 / P.g, P.g, per_wave,
		         rag ? " ragged frames" : "");
		if (two_pass) snprintf(tag16 + strlen(tag16), sizeof tag16 - strlen(tag16), " two-pass%s ck=%d", tp_split ? " (walk kernel)" : "", at::ck_steps(P.g));
		auto pick = [&](int st) {
			if (two_pass) return st == 1 ? at_pick16_tp(kmode, P.g, P.k, ts, bits, tp_split) : (at_sweep16_fn) nullptr;
			return rag ? at_pick16_rag(kmode, P.g, P.k, st, tb, bits) : at_pick16(kmode, P.g, P.k, ts, st, tb, bits);
		};
		const long long nwork = (np + per_wave - 1) / per_wave;
		/* The grid is the resident waves, each pulling work items until none are left.  A SIMD holds two of these waves and finishes
		 * an item every ~118 us: C2's 6 250 items of 16 pairs on 1 024 SIMDs are 6.1 items per SIMD, so a launch that has the chip to
		 * itself ends with 106 SIMDs working through a 7th item while the others idle (0.86 ms instead of 0.72).  When such a sliver
		 * remains (the last round less than a quarter full), its pairs become items of two 32-lane groups -- four alignments per wave,
		 * items a third as long -- that follow the main items IN THE SAME LAUNCH and work queue (the kernel's second argument,
		 * at_sweep16.hip.h).  AT_TAIL_SPLIT=0: off.  (Round 2 ran the sliver as a second launch behind the
		 * main one, which cost a caller who keeps launches in flight 8 %; round 3 first tried it on a stream of its own beside the
		 * main launch: +6.5 % alone, -8 % in flight -- the next batch's waves took the slots the sliver was waiting for.) */
		/* (the sliver's items use the main items' LDS window and pointer slot: both hold either layout -- the main items' are the larger) */
		int rc = plan_launch(h, tag16, P.k, nwork, std::max(P.off_ptr, PT.off_ptr),
		                     tp_split ? 64 : two_pass ? std::max(tpm.words, tpt.words) : std::max(P.words - P.off_ptr, PT.words - PT.off_ptr), &pl, stream,
		                     [&](int st) { return (const void *)pick(st); }, P.g < 64 || rag || two_pass);
		if (rc) return rc;
		if (tp_split) {
			/* AT_TP_RESERVE=n: the sweep leaves n wave slots per CU to the walk kernels of the launches around it (its waves are
			 * persistent: a walk kernel queued behind it otherwise waits for the whole sweep of the NEXT launch to drain) */
			const long long rsv = env_ll("AT_TP_RESERVE", 0) * h->ncu;
			if (rsv > 0 && pl.grid > rsv && nwork > pl.grid - rsv) pl.grid = std::max<long long>(h->ncu, pl.grid - rsv);
		}
		int64_t n_tail = 0;
		if (tail_ok && nwork > pl.grid) {
			const long long sliver = nwork % pl.grid;
			if (sliver > 0 && sliver * 4 <= pl.grid) n_tail = np - (nwork - sliver) * per_wave;   /* (four 32-lane items per 8-lane item: one per SIMD at most) */
		}
		Sweep16Args bt = b;
		bt.npairs = 0;
		const int64_t nm = np - n_tail;
		if (n_tail > 0) {
			b.npairs = nm;
			bt.npairs = n_tail;
			bt.woff1 += nm; bt.woff2 += nm; bt.len1 += nm; bt.len2 += nm;
			bt.score += nm;
			if (bt.end_i) bt.end_i += nm;
			if (bt.end_j) bt.end_j += nm;
			if (bt.state) bt.state += nm;
			if (bt.ops_off) bt.ops_off += nm;
			if (bt.nops) bt.nops += nm;
			bt.off_refb = PT.off_refb; bt.off_bound = PT.off_bound; bt.ptr_lanes = PT.ptr_lanes; bt.off_sm = PT.off_sm; bt.nsm = PT.nsm;
		}
		b.off_ptr = pl.off_ptr; b.ws = pl.ws; b.ws_slot_words = pl.slot_words; b.queue = h->d_queue;
		bt.off_ptr = pl.store == 1 ? 0 : PT.off_ptr; bt.ws = pl.ws; bt.ws_slot_words = pl.slot_words; bt.queue = h->d_queue;
		if (two_pass) {
			if (pl.store != 1) return fail(h, AT_ERR_RANGE, "two-pass tracebacks need the s2 windows in LDS (store=%d)", pl.store);
			b.off_brow = tpm.off_brow; b.off_rck = tpm.off_rck; b.off_cck = tpm.off_cck; b.off_rptr = tpm.off_rptr; b.off_rjpl = tpm.off_rjpl;
			bt.off_brow = tpt.off_brow; bt.off_rck = tpt.off_rck; bt.off_cck = tpt.off_cck; bt.off_rptr = tpt.off_rptr; bt.off_rjpl = tpt.off_rjpl;
		}
		if (tp_split) {
			/* [row 0 | end cells of the np alignments | the main items' checkpoints | the sliver items'] */
			const long long n_main_items = (nm + per_wave - 1) / per_wave, n_tail_items = (n_tail + 2 * (64 / at::AT_TAIL_G) - 1) / (2 * (64 / at::AT_TAIL_G));
			const long long end_words = ((np * 4) + 63) & ~63LL;
			const size_t need = (size_t)(brow_words + end_words + n_main_items * tpm.words + n_tail_items * tpt.words) * 4;
			void *pc = h->d_ck; size_t have = h->ck_bytes;
			rc = grow(h, &pc, &have, need);
			h->d_ck = (uint32_t *)pc; h->ck_bytes = have;
			if (rc && first == 0) {
				/* no room for a launch's checkpoints (several handles on one card, a small AT_CK_CAP_MB would have cut the batch into pieces):
				 * nothing has been launched yet -- the same batch with the rounds inside the sweep's kernel, whose checkpoints live in the
				 * resident wavefronts' slots */
				h->ck_alloc_failed = true;
				return align_device(h, mode, npairs, d_seq, bits, d_woff1, d_len1, d_woff2, d_len2, max_len1, max_len2, uniform_shape, want_traceback,
				                    d_score, d_end_i, d_end_j, d_state, d_ops, d_ops_off, d_nops, stream_, ap_n, ap_first, d_order, rag, only_if, only_val);
			}
			if (rc) return rc;
			b.ck_brow = h->d_ck; bt.ck_brow = h->d_ck;
			b.tp_end = (int4 *)(h->d_ck + brow_words); bt.tp_end = b.tp_end + nm;
			b.ck = h->d_ck + brow_words + end_words; b.ck_item_words = tpm.words;
			bt.ck = b.ck + n_main_items * tpm.words; bt.ck_item_words = tpt.words;
		}
		at_sweep16_fn fn16 = pick(pl.store);
		if (!fn16 || (P.g != 64 && pl.store == 2)) return fail(h, AT_ERR_RANGE, "no packed kernel for this shape (rows/lane=%d, store=%d)", P.k, pl.store);
		if (pl.dyn_lds > 48 * 1024)
			HIP_TRY(h, hipFuncSetAttribute((const void *)fn16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.dyn_lds));
		hipLaunchKernelGGL(fn16, dim3((unsigned)pl.grid), dim3(64), pl.dyn_lds, stream, b, bt);
		HIP_TRY(h, hipGetLastError());
		if (tp_split && !env_ll("AT_DIAG_NO_WALK_KERNEL", 0)) {   /* (1: throw-away runs without pass 2 -- what does the sweep alone reach?  Every pair reports garbage ops) */
			/* the 64-lane groups: teams of lanes per pair of alignments (walk16_team_wave; AT_WALK_TEAMS=0: one walker per half-lane there too) */
			const bool teams = env_ll("AT_WALK_TEAMS", P.g == 64 ? 1 : 0) && at_pick_walk16(kmode, P.g, P.k, ts, bits, 1);
			at_walk16_fn wf = at_pick_walk16(kmode, P.g, P.k, ts, bits, teams);
			if (!wf) return fail(h, AT_ERR_RANGE, "no walk kernel for %d-lane groups x %d rows", P.g, P.k);
			const size_t lds = std::max(walk_lds(P, true), tail_ok ? walk_lds(PT, true) : 0);
			if (lds > 48 * 1024) HIP_TRY(h, hipFuncSetAttribute((const void *)wf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
			/* the main units' walkers: one wavefront per 128 alignments (LDS admits four per CU; the others queue).  AT_WALK_WAVES_PER_CU = n:
			 * at most n per CU, persistent, which refill their halves from the counters behind the sweep's work counter -- fewer
			 * wavefront-rounds, but a chain of them per wavefront: C4 one launch at a time 1 450 (n = 1) / 1 820 (2) / 2 160 GCUPS (all),
			 * C2 2 300 / 2 490 / 2 500, with launches in flight 2 270 / 2 370 / 2 390 and 3 130 / 3 130 / 3 140 */
			const long long wcap = env_ll("AT_WALK_WAVES_PER_CU", 0);
			const long long wmain = std::max<long long>(1, wcap > 0 ? std::min<long long>((nm + 127) / 128, wcap * h->ncu) : (nm + 127) / 128);
			HIP_TRY(h, hipMemsetAsync(h->d_queue + 8, 0, 16, stream));
			const int tpw = 64 / at_walk16_team_lanes(P.g);   /* teams per wavefront */
			const long long wteams = ((nm + 1) / 2 + tpw - 1) / tpw;
			hipLaunchKernelGGL(wf, dim3((unsigned)((teams ? wteams : wmain) + (n_tail + 127) / 128)), dim3(64), lds, stream, b, bt);
			HIP_TRY(h, hipGetLastError());
		}
		if (n_tail > 0)
			snprintf(h->cfg + strlen(h->cfg), sizeof h->cfg - strlen(h->cfg), " + last %lld pairs as 32-lane items (rows/lane=%d)", (long long)n_tail, PT.k);
		if (first == 0) cfg_first = h->cfg;
		}
		if (piece < npairs) snprintf(h->cfg, sizeof h->cfg, "%.200s; in pieces of %lld pairs", cfg_first.c_str(), (long long)piece);
		return AT_OK;
	}

	/* ---- edit distance with unit mismatch cost: bit-parallel kernel (at_myers.hip.h), any mix of lengths ---- */
	/* (bytes of LDS for the s2 windows of the n alignments of a wavefront; two must fit, or the cell-by-cell kernel takes the batch) */
	auto myers_windows = [&](int n) { return (size_t)n * ((((size_t)max_len2 + 15) / 16 + 2) | 1) * 4; };
	if (kmode == at::K_EDIT && h->u == 1 && bits == 2 && max_len1 <= 32768 && myers_windows(2) <= 60 * 1024 && env_ll("AT_MYERS", 1)) {
		/* lanes per alignment and words per lane: reads of up to AT_MYERS_LANE_MAX (1 024) bases one alignment per LANE -- 5, 8, 16 or 32
		 * words, 64 s2 windows in LDS (second sequences of up to ~3 500 bases) -- else 32 lanes; the 16- and 32-word forms only for batches
		 * of AT_MYERS_LANE_MIN_PAIRS (16 384: one wavefront per CU) and more -- 10 000 pairs of 1 000 x 1 000 are 157 wavefronts on 1 024
		 * SIMDs, 19 TCUPS against 31 on 32-lane groups; 131 072 of them 43 against 36.  AT_MYERS_GROUP = 8: the round-2 form, eight lanes
		 * for reads of up to 256 bases and 32 beyond (A/B) */
		const long long lane_max = env_ll("AT_MYERS_LANE_MAX", 1024);
		const bool per_lane = max_len1 <= lane_max && max_len1 <= 1024 && myers_windows(64) <= 60 * 1024 &&
		                      (max_len1 <= 256 || npairs >= env_ll("AT_MYERS_LANE_MIN_PAIRS", 16384)) && env_ll("AT_MYERS_GROUP", 1) == 1;
		const int g = per_lane ? 1 : max_len1 <= 256 && myers_windows(8) <= 60 * 1024 ? 8 : 32;
		const int w = per_lane ? (max_len1 <= 64 ? 2 : max_len1 <= 96 ? 3 : max_len1 <= 128 ? 4 : max_len1 <= 160 ? 5 : max_len1 <= 256 ? 8 : max_len1 <= 512 ? 16 : 32)
		            : max_len1 <= 1024 ? 1 : max_len1 <= 2048 ? 2 : max_len1 <= 4096 ? 4 : max_len1 <= 8192 ? 8 : max_len1 <= 16384 ? 16 : 32;
		at_myers_fn fn = at_pick_myers(w, g);
		const int per_wave = 64 / g;
		at::MyersArgs m;
		memset(&m, 0, sizeof m);
		m.npairs = npairs; m.seq = d_seq;
		m.woff1 = (const long long *)d_woff1; m.woff2 = (const long long *)d_woff2; m.len1 = d_len1; m.len2 = d_len2;
		m.max_l1 = max_len1; m.max_l2 = max_len2;
		m.score = d_score; m.end_i = d_end_i; m.end_j = d_end_j; m.state = d_state; m.nops = d_nops;
		m.order = d_order;
		m.ap_n = ap_n; m.ap_first = ap_first;
		if (!h->d_queue) HIP_TRY(h, hipMalloc((void **)&h->d_queue, 128));
		HIP_TRY(h, hipMemsetAsync(h->d_queue, 0, 8, stream));
		m.queue = h->d_queue;
		const size_t lds = myers_windows(per_wave);   /* (odd window stride, as the kernel computes it) */
		if (lds > 60 * 1024) return fail(h, AT_ERR_RANGE, "second sequence too long for the bit-parallel kernel's LDS window");
		int occ = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)fn, 64, lds) != hipSuccess || occ <= 0) occ = 8;
		const long long grid = std::max(1LL, std::min<long long>((npairs + per_wave - 1) / per_wave, (long long)occ * h->ncu));
		hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(64), lds, stream, m);
		HIP_TRY(h, hipGetLastError());
		snprintf(h->cfg, sizeof h->cfg, "myers bits=2 words/lane=%d %dx%d-lane groups (%d pairs/wave) lds=%zuB waves/cu<=%d grid=%lld", w, per_wave, g, per_wave,
		         lds, occ, grid);
		return AT_OK;
	}

	/* ---- all-vs-all overlap scores with a threshold (at_set_min_score): the bit-parallel filter (at_myers.hip.h, SEMI) bounds every
	 * pair's score from above at a seventh of the sweep's cost; only pairs whose bound reaches the threshold are swept.  The filter
	 * leaves (upper bound, state 0) in every pair's result slot and the candidates' indices in a list; the sweep below takes the
	 * list and its length from the device and overwrites the candidates' slots with exact results.  No host round trip. ---- */
	const int *npairs_dev = nullptr;
	char filter_note[96] = "";
	if (kmode == at::K_OVERLAP && !tb && ap_n > 0 && h->min_on && !d_order && !only_if && bits == 2 && max_len1 >= 1 && max_len1 <= 1024 &&
	    npairs < (1LL << 31) - 64 && env_ll("AT_OVERLAP_FILTER", 1)) {
		/* 2 score <= 2 m b - (2 c - m) D', c = min(m - u, m / 2 - o): needs m >= 0 and 2 c > m */
		const long long k2 = std::min<long long>(2LL * (h->m - h->u), (long long)h->m - 2LL * h->o) - h->m;
		auto windows = [&](int n) { return (size_t)n * ((((size_t)max_len2 + 15) / 16 + 2) | 1) * 4; };
		const int w = max_len1 <= 64 ? 2 : max_len1 <= 96 ? 3 : max_len1 <= 128 ? 4 : max_len1 <= 160 ? 5 : max_len1 <= 256 ? 8 : max_len1 <= 512 ? 16 : 32;
		at_myers_fn ffn = at_pick_myers_semi(w);
		if (h->m >= 0 && k2 > 0 && k2 < 4096 && h->m < 4096 && windows(64) <= 60 * 1024 && ffn) {
			int rc = grow(h, &h->d_order, &h->order_bytes, ((size_t)npairs + 64) * 4);
			if (rc) return rc;
			int *cand = (int *)h->d_order, *count = cand + ((npairs + 15) & ~15LL);
			HIP_TRY(h, hipMemsetAsync(count, 0, 4, stream));
			at::MyersArgs m;
			memset(&m, 0, sizeof m);
			m.npairs = npairs; m.seq = d_seq;
			m.woff1 = (const long long *)d_woff1; m.woff2 = (const long long *)d_woff2; m.len1 = d_len1; m.len2 = d_len2;
			m.max_l1 = max_len1; m.max_l2 = max_len2;
			m.score = d_score; m.end_i = d_end_i; m.end_j = d_end_j; m.state = d_state;
			m.ap_n = ap_n; m.ap_first = ap_first;
			m.semi_m2 = 2 * h->m; m.semi_k = (int)k2; m.semi_min2 = (int)std::max<long long>(std::min<long long>(2LL * h->min_score, INT32_MAX), INT32_MIN);
			m.cand_order = cand; m.cand_count = count;
			if (!h->d_queue) { HIP_TRY(h, hipMalloc((void **)&h->d_queue, 128)); HIP_TRY(h, hipMemset(h->d_queue, 0, 128)); }
			HIP_TRY(h, hipMemsetAsync(h->d_queue, 0, 8, stream));
			m.queue = h->d_queue;
			const size_t lds = windows(64);
			int occ = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)ffn, 64, lds) != hipSuccess || occ <= 0) occ = 8;
			const long long fgrid = std::max(1LL, std::min<long long>((npairs + 63) / 64, (long long)occ * h->ncu));
			hipLaunchKernelGGL(ffn, dim3((unsigned)fgrid), dim3(64), lds, stream, m);
			HIP_TRY(h, hipGetLastError());
			d_order = cand; npairs_dev = count;
			snprintf(filter_note, sizeof filter_note, "overlap filter (bit-parallel bound, %d words/lane, min score %d) + ", w, h->min_score);
		}
	}

	/* the int32 kernels read 2-bit scores from a byte LUT (the packed ones above from a 16-bit one) */
	if (bits == 2 && !scores_fit_byte(h, mode))
		return fail(h, AT_ERR_RANGE, "the int32 2-bit kernels need |16*score| <= 127 (m=%d u=%d o=%d): pack the batch with bits=8", h->m, h->u, h->o);
	const Layout L = layout_for(kmode, bits, tb, max_len1, max_len2);
	if (L.words >= (1LL << 30)) return fail(h, AT_ERR_RANGE, "pair too large: %lld workspace words", L.words);
	SweepArgs a;
	memset(&a, 0, sizeof a);
	a.npairs = npairs;
	a.seq = d_seq;
	a.woff1 = (const long long *)d_woff1; a.len1 = d_len1;
	a.woff2 = (const long long *)d_woff2; a.len2 = d_len2;
	a.m16 = h->m * 16; a.u16 = h->u * 16; a.o16 = h->o * 16; a.e16 = h->e * 16; a.g16 = h->j * 16;
	a.u_raw = h->u;
	a.score = d_score; a.end_i = d_end_i; a.end_j = d_end_j; a.state = d_state;
	a.ops = d_ops; a.ops_off = (const long long *)d_ops_off; a.nops = d_nops;
	a.off_bound = L.off_bound; a.ptr_lanes = L.ptr_lanes; a.off_sm = L.off_sm; a.nsm = L.nsm;
	a.ap_n = ap_n; a.ap_first = ap_first; a.order = d_order;
	a.only_if = only_if; a.only_val = only_val;
	a.npairs_dev = npairs_dev;
	if (kmode == at::K_FITJ) {
		int rc = ensure_sitemask(h, max_len2, stream);
		if (rc) return rc;
		a.sitemask = h->d_sitemask;
	}
	Plan pl;
	char tag[32];
	snprintf(tag, sizeof tag, "int32 bits=%d", bits);
	int rc = plan_launch(h, tag, L.k, npairs, L.off_ptr, L.words - L.off_ptr, &pl, stream, [&](int st) {
		return (const void *)(bits == 2 ? at_pick32_b2(kmode, L.k, st, tb) : at_pick32_b8(kmode, L.k, st, tb));
	});
	if (rc) return rc;
	a.off_ptr = pl.off_ptr; a.ws = pl.ws; a.ws_slot_words = pl.slot_words; a.queue = h->d_queue;
	a.max_l1 = max_len1; a.max_l2 = max_len2;
	at_sweep_fn fn = bits == 2 ? at_pick32_b2(kmode, L.k, pl.store, tb) : at_pick32_b8(kmode, L.k, pl.store, tb);
	if (pl.dyn_lds > 48 * 1024)
		HIP_TRY(h, hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.dyn_lds));
	hipLaunchKernelGGL(fn, dim3((unsigned)pl.grid), dim3(64), pl.dyn_lds, stream, a);
	HIP_TRY(h, hipGetLastError());
	if (filter_note[0]) {
		const std::string tail = h->cfg;
		snprintf(h->cfg, sizeof h->cfg, "%s%.200s", filter_note, tail.c_str());
	}
	return AT_OK;
}

/* ---- host entry: 2-bit packing while staging (round 4) ----
 * The host entry used to copy every raw byte into page-locked memory and send it up (30 MB per 100k pairs of C2), the GPU packed.
 * The staging pass already touches every byte: it now packs them -- 16 bases per word, the layout of at_pack<2> -- so a quarter of
 * the bytes cross the link, and the sweeps, which wait for their chunk's upload, start that much earlier.  32 bases per step:
 * code = ((c >> 1) ^ (c >> 2)) & 3 maps A C G T to 0 1 2 3, a byte shuffle of "ACGT" by the code must give the byte back (else the
 * batch is not pure ACGT: the raw path takes over, as before), two multiply-adds fold four codes into a byte. */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static inline __m256i pack2_step(__m256i v, __m256i bad, uint32_t *dst)
{
	const __m256i m3 = _mm256_set1_epi8(3), w14 = _mm256_set1_epi16(0x0401), w116 = _mm256_set1_epi32(0x00100001);
	const __m256i acgt = _mm256_setr_epi8('A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'G', 'T', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
	const __m256i pickb = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
	const __m256i code = _mm256_and_si256(_mm256_xor_si256(_mm256_srli_epi16(v, 1), _mm256_srli_epi16(v, 2)), m3);
	const __m256i x = _mm256_madd_epi16(_mm256_maddubs_epi16(code, w14), w116);   /* a byte per 4 bases in every dword */
	const __m256i y = _mm256_shuffle_epi8(x, pickb);
	const uint64_t r = (uint64_t)(uint32_t)_mm256_extract_epi32(y, 0) | ((uint64_t)(uint32_t)_mm256_extract_epi32(y, 4) << 32);
	memcpy(dst, &r, 8);
	return _mm256_or_si256(bad, _mm256_xor_si256(_mm256_shuffle_epi8(acgt, code), v));
}
/* `len` bytes of ACGT -> (len + 15) / 16 + 1 words (the last one the padding word the kernels' windows read ahead into); false if a
 * byte is not one of ACGT (the words are then unspecified) */
__attribute__((target("avx2"))) static bool pack2_avx2(const uint8_t *src, int len, uint32_t *dst)
{
	const int nw = (len + 15) / 16 + 1;
	__m256i bad = _mm256_setzero_si256();
	int b = 0;
	for (; b + 32 <= len; b += 32) bad = pack2_step(_mm256_loadu_si256((const __m256i *)(src + b)), bad, dst + b / 16);
	if (b < len) {   /* (the last 1 .. 31 bases, filled up with A = code 0: its two words end at or before the padding word) */
		alignas(32) uint8_t tail[32];
		memset(tail, 'A', 32);
		memcpy(tail, src + b, (size_t)(len - b));
		bad = pack2_step(_mm256_load_si256((const __m256i *)tail), bad, dst + b / 16);
	}
	dst[nw - 1] = 0;
	return _mm256_testz_si256(bad, bad) != 0;
}
static bool host_pack_available() { static const bool ok = __builtin_cpu_supports("avx2"); return ok; }
#else
static bool pack2_avx2(const uint8_t *, int, uint32_t *) { return false; }
static bool host_pack_available() { return false; }
#endif

/* AT_HOST_TRACE=1: microseconds since the first call at the stages of the host entry, on stderr (where does a call's time go?) */
static void htrace(const char *what, long long a)
{
	static const bool on = getenv("AT_HOST_TRACE") != nullptr;
	if (!on) return;
	static const auto t0 = std::chrono::steady_clock::now();
	const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
	fprintf(stderr, "[host %10.1f us] %s %lld\n", us, what, a);
}

/* host-buffer entry; with out_r1/out_r2 the strings are rendered on the GPU (slots of len1+len2+1 bytes at ops_off[k])
 * and the op codes stay on the device */
static int align_host(at_handle *h, int mode, int64_t npairs, const uint8_t *seq_blob,
                      const int64_t *off1, const int32_t *len1, const int64_t *off2, const int32_t *len2,
                      int want_traceback,
                      int32_t *out_score, int32_t *out_end_i, int32_t *out_end_j, int32_t *out_state,
                      uint8_t *out_ops, const int64_t *ops_off, int32_t *out_nops, char *out_r1, char *out_r2,
                      int64_t pair_base = 0)   /* index of pair 0 in the caller's batch, for messages */
{
	if (!h) return fail(nullptr, AT_ERR_ARG, "at_align_batch: NULL handle");
	if (mode < AT_MODE_GLOBAL || mode > AT_MODE_EDIT) return fail(h, AT_ERR_ARG, "unknown mode %d", mode);
	if (npairs < 0) return fail(h, AT_ERR_ARG, "negative npairs");
	if (npairs == 0) return AT_OK;
	if (!seq_blob || !off1 || !len1 || !off2 || !len2 || !out_score) return fail(h, AT_ERR_ARG, "NULL argument");
	const bool tb = want_traceback && mode != AT_MODE_EDIT;
	const bool strings = out_r1 != nullptr;
	if (tb && ((!out_ops && !strings) || !ops_off || !out_nops)) return fail(h, AT_ERR_ARG, "traceback wanted but ops buffers are NULL");

	int max1 = 0, max2 = 0;
	bool uniform = true;
	int64_t ops_total = 0, ops_lo = INT64_MAX, blob_lo = INT64_MAX, slots_total = 0;
	for (int64_t k = 0; k < npairs; ++k) {
		if (len1[k] < 0 || len2[k] < 0) return fail(h, AT_ERR_ARG, "pair %lld: negative length", (long long)(pair_base + k));
		/* the domain on which the reference is defined (last paragraph) */
		if (mode == AT_MODE_FIT && len1[k] > len2[k])
			return fail(h, AT_ERR_FIT_ORDER, "first sequence must be shorter than the second");
		if (mode == AT_MODE_LOCAL && (len1[k] < 1 || len2[k] < 1)) return fail(h, AT_ERR_DOMAIN, "pair %lld: local needs non-empty sequences", (long long)(pair_base + k));
		if (mode == AT_MODE_FIT && len1[k] < 1) return fail(h, AT_ERR_DOMAIN, "pair %lld: fit needs a non-empty read", (long long)(pair_base + k));
		if (mode == AT_MODE_OVERLAP && len2[k] < 1) return fail(h, AT_ERR_DOMAIN, "pair %lld: overlap needs a non-empty second sequence", (long long)(pair_base + k));
		max1 = std::max(max1, len1[k]); max2 = std::max(max2, len2[k]);
		if (len1[k] != len1[0] || len2[k] != len2[0]) uniform = false;
		if (tb) {
			if (ops_off[k] < 0) return fail(h, AT_ERR_ARG, "pair %lld: negative ops offset", (long long)(pair_base + k));
			ops_total = std::max<int64_t>(ops_total, ops_off[k] + len1[k] + len2[k] + (strings ? 1 : 0));
			ops_lo = std::min<int64_t>(ops_lo, ops_off[k]);
			slots_total += (int64_t)len1[k] + len2[k];
		}
		if (off1[k] < 0 || off2[k] < 0) return fail(h, AT_ERR_ARG, "pair %lld: negative sequence offset", (long long)(pair_base + k));
		blob_lo = std::min<int64_t>(blob_lo, std::min(off1[k], off2[k]));
	}
	htrace("chunk: lengths checked, pair", pair_base);
	/* only the span of the blob and of the ops buffer that this call touches travels (a chunk of a larger batch,
	 * see at_align_batch, addresses the caller's buffers with absolute offsets) */
	if (!tb) ops_lo = 0;
	ops_total -= ops_lo;
	HIP_TRY(h, hipSetDevice(h->device));

	/* ---- inputs go up RAW; packing happens on the GPU (at_pack.hip.h).  Host work is O(npairs): one pass writes every descriptor
	 * ---- array into ONE page-locked block that mirrors the device block, so all of them travel in one copy; the raw bytes are
	 * ---- staged through page-locked memory in pieces by this thread (the chunks of a batch run on threads of their own: their
	 * ---- staging copies run side by side, where the runtime's own pageable path took them one after the other) ---- */
	auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t n = (size_t)npairs;
	/* descriptor block, uploaded: soff[2n] swoff[2n] opsoff[n] (int64) slen[2n] (int32) -- per SEQUENCE, s1 of pair k at 2k, s2 at 2k + 1;
	 * behind it, made on the device (at_split_desc): woff1[n] woff2[n] (int64) len1[n] len2[n] (int32), what the sweep kernels take.
	 * The word offsets for byte words wait in page-locked memory in case a batch turns out not to be pure ACGT. */
	const size_t o_soff = 0, o_sw = o_soff + al(2 * n * 8), o_ops = o_sw + al(2 * n * 8), o_slen = o_ops + al(n * 8), up_bytes = o_slen + al(2 * n * 4);
	const size_t o_w1 = up_bytes, o_w2 = o_w1 + al(n * 8), o_l1 = o_w2 + al(n * 8), o_l2 = o_l1 + al(n * 4), desc_bytes = o_l2 + al(n * 4);
	int rc = grow_pinned(h, &h->hp_desc, &h->hp_desc_bytes, up_bytes + al(2 * n * 8));
	if (rc) return rc;
	char *hd = (char *)h->hp_desc;
	int64_t *p_soff = (int64_t *)(hd + o_soff), *p_sw = (int64_t *)(hd + o_sw), *p_ops = (int64_t *)(hd + o_ops), *p_sw8 = (int64_t *)(hd + up_bytes);
	int32_t *p_slen = (int32_t *)(hd + o_slen);
	const bool force8 = !scores_fit_byte(h, mode);    /* large scores: byte-compare kernels */
	int64_t nwords2 = 0, nwords8 = 0, blob_bytes = 0;
	for (int64_t k = 0; k < npairs; ++k) {
		const int a1 = len1[k], a2 = len2[k];
		p_soff[2 * k] = off1[k] - blob_lo; p_soff[2 * k + 1] = off2[k] - blob_lo;
		p_slen[2 * k] = a1; p_slen[2 * k + 1] = a2;
		p_sw[2 * k] = nwords2; nwords2 += (a1 + 15) / 16 + 1;
		p_sw[2 * k + 1] = nwords2; nwords2 += (a2 + 15) / 16 + 1;
		p_sw8[2 * k] = nwords8; nwords8 += (a1 + 3) / 4 + 1;
		p_sw8[2 * k + 1] = nwords8; nwords8 += (a2 + 3) / 4 + 1;
		blob_bytes = std::max<int64_t>(blob_bytes, std::max(p_soff[2 * k] + a1, p_soff[2 * k + 1] + a2));
		if (tb) p_ops[k] = ops_off[k] - ops_lo;
	}
	if (force8) memcpy(p_sw, p_sw8, 2 * n * 8);
	const int64_t nwords_max = std::max(nwords2, nwords8) + 4;
	/* device input block: words | descriptor block | flag | raw blob */
	const size_t b_words = al((size_t)nwords_max * 4), b_blob = al((size_t)blob_bytes + 32);   /* (at_pack reads whole dwords: up to 20 bytes behind the last base) */
	rc = grow(h, &h->d_in, &h->in_bytes, b_words + desc_bytes + 256 + b_blob);
	if (rc) return rc;
	char *din = (char *)h->d_in, *dd = din + b_words;
	uint32_t *d_words = (uint32_t *)din;
	int64_t *d_soff = (int64_t *)(dd + o_soff), *d_swoff = (int64_t *)(dd + o_sw), *d_opsoff = (int64_t *)(dd + o_ops);
	int64_t *d_woff1 = (int64_t *)(dd + o_w1), *d_woff2 = (int64_t *)(dd + o_w2);
	int32_t *d_slen = (int32_t *)(dd + o_slen), *d_len1 = (int32_t *)(dd + o_l1), *d_len2 = (int32_t *)(dd + o_l2);
	int *d_flag = (int *)(dd + desc_bytes);
	uint8_t *d_blob = (uint8_t *)(dd + desc_bytes + 256);
	hipStream_t s = h->stream;
	/* a caller whose sequences already lie in page-locked memory (hipHostMalloc / hipHostRegister) is copied from in place */
	bool caller_pinned = false;
	{
		hipPointerAttribute_t pattr;
		if (hipPointerGetAttributes(&pattr, seq_blob + blob_lo) == hipSuccess) caller_pinned = pattr.type == hipMemoryTypeHost;
		else (void)hipGetLastError();                  /* (ordinary memory: not an error) */
		if (env_ll("AT_HOST_NO_PINNED_CALLER", 0)) caller_pinned = false;
	}
	const uint8_t *up_src = seq_blob + blob_lo;
	if (!caller_pinned) {
		rc = grow_pinned(h, &h->hp_blob, &h->hp_blob_bytes, (size_t)blob_bytes + 64);
		if (rc) return rc;
	}
	/* (Tried and dropped, twice: the chunks of a batch uploading in order -- own streams chained by events, or one copy stream with
	 * the threads waiting on the host for their own copies -- so that each crosses the link at its full rate and the first sweep
	 * starts early.  Round 2: 2.28 / 2.37 ms per 100k pairs of C2 against 2.17 side by side.  Round 3, medians of 50 calls on one box:
	 * 1.90-2.12 ms against 1.99-2.04 with tracebacks, and 1.65-1.77 against 1.21-1.37 scores only -- copies queued side by side on six
	 * streams move more bytes per second than the same copies one behind the other.) */
	HIP_TRY(h, hipMemcpyAsync(dd, hd, up_bytes, hipMemcpyHostToDevice, s));
	/* pure-ACGT batches are packed HERE, while they are staged: the words go up (a quarter of the bytes), the raw blob and the GPU's
	 * packing kernel are skipped.  The first byte that is not ACGT ends the attempt: the raw path below takes the whole chunk. */
	bool host_packed = false;
	if (!force8 && !caller_pinned && host_pack_available() && env_ll("AT_HOST_PACK", 1)) {
		rc = grow_pinned(h, &h->hp_blob, &h->hp_blob_bytes, (size_t)(nwords2 + 4) * 4 + 64);
		if (rc) return rc;
		uint32_t *hw = (uint32_t *)h->hp_blob;
		const int64_t piece_words = std::max<long long>(4096, env_ll("AT_HOST_STAGE_PIECE", 4 << 20)) / 16;   /* (a quarter of the raw pieces' bytes) */
		int64_t sent = 0;
		bool ok = true;
		for (int64_t k = 0; k < 2 * npairs && ok; ++k) {
			ok = pack2_avx2(up_src + p_soff[k], p_slen[k], hw + p_sw[k]);
			const int64_t done = k + 1 < 2 * npairs ? p_sw[k + 1] : nwords2;
			if (ok && (done - sent >= piece_words || k + 1 == 2 * npairs)) {
				if (k + 1 == 2 * npairs) for (int x = 0; x < 4; ++x) hw[nwords2 + x] = 0;
				const int64_t upto = k + 1 == 2 * npairs ? nwords2 + 4 : done;
				HIP_TRY(h, hipMemcpyAsync(d_words + sent, hw + sent, (size_t)(upto - sent) * 4, hipMemcpyHostToDevice, s));
				sent = upto;
			}
		}
		host_packed = ok;
		if (!ok) HIP_TRY(h, hipStreamSynchronize(s));     /* (hp_blob is about to be reused for the raw bytes) */
	}
	if (host_packed) {
	} else if (caller_pinned) HIP_TRY(h, hipMemcpyAsync(d_blob, up_src, (size_t)blob_bytes, hipMemcpyHostToDevice, s));
	else {
		/* pieces of 4 MB: the staging of one beside the transfer of the one before (copies of 1 MB cross the link at 32 GB/s,
		 * of 5 MB at 50: tools/pcie_rate.py) */
		const size_t piece = (size_t)std::max<long long>(4096, env_ll("AT_HOST_STAGE_PIECE", 4 << 20));
		for (size_t at = 0; at < (size_t)blob_bytes; at += piece) {
			const size_t nb = std::min(piece, (size_t)blob_bytes - at);
			memcpy((char *)h->hp_blob + at, up_src + at, nb);
			HIP_TRY(h, hipMemcpyAsync(d_blob + at, (char *)h->hp_blob + at, nb, hipMemcpyHostToDevice, s));
		}
	}
	at::PackArgs pa;
	pa.nseq = 2 * npairs; pa.blob = d_blob; pa.off = (const long long *)d_soff; pa.len = d_slen;
	pa.woff = (const long long *)d_swoff; pa.words = d_words; pa.not_acgt = d_flag;
	const unsigned pgrid = (unsigned)std::min<int64_t>((2 * npairs + 15) / 16, 8LL * h->ncu);
	int bits = force8 ? 8 : 2;
	int *p_flag = (int *)((char *)h->hp_flag);
	bool flag_pending = false;
	if (bits == 2 && !host_packed) {
		HIP_TRY(h, hipMemsetAsync(d_flag, 0, 4, s));
		hipLaunchKernelGGL(at::at_pack<2>, dim3(pgrid), dim3(256), 0, s, pa);
		HIP_TRY(h, hipMemcpyAsync(p_flag, d_flag, 4, hipMemcpyDeviceToHost, s));
		flag_pending = true;
	}
	htrace("chunk: uploads and packing queued, pair", pair_base);
	/* The alphabet of the batch is known when the packing kernel's flag is down -- one wait for the uploads per chunk.  A ragged
	 * batch makes its plan (frames, the order of the pairs: host work over the lengths alone) BEFORE that wait, for the 2-bit
	 * alphabet it expects, and again only if the flag says otherwise. */
	auto settle_alphabet = [&]() -> int {
		if (flag_pending) {
			HIP_TRY(h, hipStreamSynchronize(s));
			htrace("chunk: packed, pair", pair_base);
			flag_pending = false;
			if (*p_flag) {                                 /* some byte is not one of ACGT: byte words, byte kernels */
				bits = 8;
				HIP_TRY(h, hipMemcpyAsync(d_swoff, p_sw8, 2 * n * 8, hipMemcpyHostToDevice, s));
			}
		}
		if (bits == 8) hipLaunchKernelGGL(at::at_pack<8>, dim3(pgrid), dim3(256), 0, s, pa);
		hipLaunchKernelGGL(at::at_split_desc, dim3((unsigned)std::min<int64_t>((npairs + 255) / 256, 8LL * h->ncu)), dim3(256), 0, s,
		                   (const long long *)d_swoff, (const int *)d_slen, (long long)npairs, (long long *)d_woff1, (long long *)d_woff2, d_len1, d_len2);
		HIP_TRY(h, hipGetLastError());
		return AT_OK;
	};
	const size_t b_len1 = al((size_t)npairs * 4);
	/* device output block: score | end_i | end_j | state | nops | ops */
	const size_t b_ops = al((size_t)ops_total + 64), b_pfx = al((size_t)(npairs + 1) * 8);
	const size_t out_need = 5 * b_len1 + b_ops + (tb ? b_pfx : 0);
	rc = grow(h, &h->d_out, &h->out_bytes, out_need);
	if (rc) return rc;
	char *dout = (char *)h->d_out;
	int32_t *d_score = (int32_t *)dout, *d_ei = (int32_t *)(dout + b_len1), *d_ej = (int32_t *)(dout + 2 * b_len1);
	int32_t *d_st = (int32_t *)(dout + 3 * b_len1), *d_nops = (int32_t *)(dout + 4 * b_len1);
	uint8_t *d_ops = (uint8_t *)(dout + 5 * b_len1);
	int64_t *d_poff = (int64_t *)(dout + 5 * b_len1 + b_ops);   /* exclusive prefix sums of nops (tb only) */

	/* ragged batch.  Affine alignments of reads (l1 <= 304, scores within 16 bits) go to the packed kernels in FRAMES
	 * (RAG kernels): a work item sweeps the extents its alignments need and every alignment keeps its own.
	 *   local          pairs sorted by (rows-per-lane class of l1, l2), cut into buckets of similar l2, one launch per
	 *                  bucket on the 16-lane groups; the alignments of an item may differ in l1 and l2
	 *   global / fit   their end cells lie in row l1, so the alignments of an item share l1: pairs sorted by (l1, l2), every
	 *                  run of equal l1 padded to whole work items by repeating its last pair (which is then computed twice,
	 *                  same result to the same place), one launch per rows-per-lane class -- 8-lane groups up to 152 bases,
	 *                  16-lane groups up to 304; an item sweeps the largest l2 among its own alignments
	 * Everything else: the int32 kernel, pairs handed out largest first (the work queue is dynamic, so a big pair
	 * picked up last would otherwise run alone at the end). */
	int *d_order = nullptr;
	bool frames = false;
	std::vector<int> order;
	const bool ragged = !uniform && npairs > 1 && npairs < (1LL << 31);
	if (!ragged) {
		rc = settle_alphabet();
		if (rc) return rc;
	}
	if (ragged) {
		int th = 0, min1 = INT32_MAX, min2 = INT32_MAX;
		for (int64_t k = 0; k < npairs; ++k) { min1 = std::min(min1, len1[k]); min2 = std::min(min2, len2[k]); }
		const bool affine = mode == AT_MODE_LOCAL || mode == AT_MODE_GLOBAL || mode == AT_MODE_FIT;
		const bool ovl = mode == AT_MODE_OVERLAP && tb;
		const int kmode_f = mode == AT_MODE_LOCAL ? at::K_LOCAL : mode == AT_MODE_GLOBAL ? at::K_GLOBAL : mode == AT_MODE_OVERLAP ? at::K_OVERLAP
		                  : h->use_jump ? at::K_FITJ : at::K_FIT;
		/* the longest read with a one-strip frame: 32 lanes x 19 rows for local, x 16 for global, x 13 for fit (the uniform kernels'
		 * classes, layout16_for); overlap: 64 lanes x 16 rows */
		const int max_rag = mode == AT_MODE_LOCAL ? 608 : mode == AT_MODE_GLOBAL ? 512 : mode == AT_MODE_FIT ? 416 : 1024;
		/* (group width, rows per lane) of a read length.  Local frames mix read lengths freely and run on the 16-lane groups up to
		 * 304 bases (on the 8-lane groups, whose lanes carry up to 19 rows, the same batches ran 15 % slower: 100..150 x 100..150 2.9
		 * against 2.5 ms per 100k pairs), on the 32-lane groups beyond; global / fit: 8-lane groups up to 152 bases, 16-lane up to
		 * 304, 32-lane beyond; overlap: the 64-lane group with 4 rows per lane up to 256 bases, 16 beyond */
		auto gclass = [&](int l1) { return ovl ? 64 : l1 > 304 ? 32 : mode == AT_MODE_LOCAL ? 16 : l1 <= 152 ? 8 : 16; };
		auto kclass = [&](int l1) {
			if (ovl) return l1 <= 256 ? 4 : 16;
			if (l1 > 304) return l1 <= 320 ? 10 : l1 <= 384 ? 12 : l1 <= 416 ? 13 : l1 <= 512 ? 16 : 19;
			if (mode == AT_MODE_LOCAL) return l1 <= 64 ? 4 : l1 <= 80 ? 5 : l1 <= 96 ? 6 : l1 <= 112 ? 7 : l1 <= 160 ? 10 : l1 <= 208 ? 13 : l1 <= 256 ? 16 : 19;
			return l1 <= 40 ? 5 : l1 <= 48 ? 6 : l1 <= 56 ? 7 : l1 <= 64 ? 8 : l1 <= 80 ? 10 : l1 <= 104 ? 13 : l1 <= 128 ? 16 : l1 <= 152 ? 19 : l1 <= 160 ? 10 : l1 <= 208 ? 13 : l1 <= 256 ? 16 : 19;
		};
		for (int pass = 0; pass < 2; ++pass) {   /* (the second pass: the batch was not pure ACGT after all) */
			const int planned_bits = bits;
			order.resize((size_t)npairs);
			for (int64_t k = 0; k < npairs; ++k) order[(size_t)k] = (int)k;
			frames = (affine || ovl) && max1 <= max_rag && min1 >= 1 && min2 >= 1 && npairs >= 64 && env_ll("AT_RAGGED_PACKED", 1) &&
			         packed_ok(h, mode, bits, max1, max2, ovl ? 2 : 4, &th);
			/* every frame is at most max1 x max2: if a class has no packed kernel for that (s2 too long for LDS), none is tried */
			if (frames) {
				const bool hasj = kmode_f == at::K_FITJ;
				const int tsf = ovl ? 2 : 4;
				const int tops[4] = {std::min(max1, 152), std::min(max1, 304), std::min(max1, 608), max1};
				for (int q = 0; q < 4 && frames; ++q) {
					const int l1q = ovl ? max1 : tops[q];
					if (l1q < min1 || (q > 0 && !ovl && tops[q] == tops[q - 1])) continue;
					const int g = gclass(l1q);
					frames = packed16_kernel_exists(kmode_f, layout16_for(tb, hasj, l1q, max2, tsf, g, ovl, kmode_f, ovl ? kclass(l1q) : 0), tb, tsf, bits, g);
					if (ovl) break;
				}
			}
			if (frames && mode == AT_MODE_LOCAL) {
				/* (class descending, l2 descending, index ascending): a counting sort -- the key space is 13 x (max2 + 1) */
				auto kidx = [&](int l1) {   /* classes in descending order of rows */
					if (l1 > 304) { const int kc = kclass(l1); return kc == 19 ? 0 : kc == 16 ? 1 : kc == 13 ? 2 : kc == 12 ? 3 : 4; }
					const int kc = kclass(l1);
					return 5 + (kc == 19 ? 0 : kc == 16 ? 1 : kc == 13 ? 2 : kc == 10 ? 3 : kc == 7 ? 4 : kc == 6 ? 5 : kc == 5 ? 6 : 7);
				};
				const size_t span = (size_t)max2 + 1;
				std::vector<int> start(13 * span + 1, 0);
				for (int64_t k = 0; k < npairs; ++k) ++start[(size_t)kidx(len1[k]) * span + (size_t)(max2 - len2[k]) + 1];
				for (size_t q = 1; q < start.size(); ++q) start[q] += start[q - 1];
				for (int64_t k = 0; k < npairs; ++k) order[(size_t)start[(size_t)kidx(len1[k]) * span + (size_t)(max2 - len2[k])]++] = (int)k;
			} else if (frames) {
				/* (l1 descending, l2 descending, index ascending) by counting sort, then every run of equal l1 padded to whole
				 * work items (16 alignments on the 8-lane groups, 8 on the 16-lane groups, 4 on the 32-lane groups, 2 on the 64-lane group) */
				const size_t span = (size_t)max2 + 1;
				std::vector<int> start((size_t)(max1 + 1) * span + 1, 0);
				for (int64_t k = 0; k < npairs; ++k) ++start[(size_t)(max1 - len1[k]) * span + (size_t)(max2 - len2[k]) + 1];
				for (size_t q = 1; q < start.size(); ++q) start[q] += start[q - 1];
				std::vector<int> sorted((size_t)npairs);
				for (int64_t k = 0; k < npairs; ++k) sorted[(size_t)start[(size_t)(max1 - len1[k]) * span + (size_t)(max2 - len2[k])]++] = (int)k;
				order.clear();
				for (size_t b0 = 0; b0 < sorted.size();) {
					size_t b1 = b0;
					const size_t run0 = order.size();
					while (b1 < sorted.size() && len1[sorted[b1]] == len1[sorted[b0]]) order.push_back(sorted[b1++]);
					const size_t per = (size_t)(2 * (64 / gclass(len1[sorted[b0]])));
					while ((order.size() - run0) % per) order.push_back(~sorted[b1 - 1]);   /* ~index: swept, not stored (at_sweep16.hip.h) */
					b0 = b1;
				}
			} else
				std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
					return (int64_t)len1[x] * len2[x] > (int64_t)len1[y] * len2[y];
				});
			if (pass == 0) {
				rc = settle_alphabet();
				if (rc) return rc;
			}
			if (bits == planned_bits) break;
		}
		htrace("chunk: planned, pair", pair_base);
		/* the order goes up from page-locked memory: nothing to wait for on the host */
		rc = grow(h, &h->d_order, &h->order_bytes, order.size() * 4);
		if (rc) return rc;
		rc = grow_pinned(h, &h->hp_order, &h->hp_order_bytes, order.size() * 4);
		if (rc) return rc;
		memcpy(h->hp_order, order.data(), order.size() * 4);
		d_order = (int *)h->d_order;
		HIP_TRY(h, hipMemcpyAsync(d_order, h->hp_order, order.size() * 4, hipMemcpyHostToDevice, s));
		if (frames && mode == AT_MODE_LOCAL) {
			/* (a bucket is a launch, and the launches of a chunk follow each other on its stream: 4 096 pairs per bucket and 33k-pair chunks
			 * made up to eight launches per class -- 30..150-base reads 2.24 ms per 100k pairs, 1.92-1.96 with at most two) */
			const int64_t min_bucket = env_ll("AT_RAGGED_MIN_BUCKET", 16384);
			int nb = 0;
			for (int64_t b0 = 0; b0 < npairs;) {
				const int g = gclass(len1[order[(size_t)b0]]), kc = kclass(len1[order[(size_t)b0]]);
				const int l2first = len2[order[(size_t)b0]];
				int64_t b1 = b0;
				int f1 = 0;
				while (b1 < npairs) {
					const int x = order[(size_t)b1];
					if (kclass(len1[x]) != kc || gclass(len1[x]) != g) break;
					if (b1 - b0 >= min_bucket && (int64_t)len2[x] * 5 < (int64_t)l2first * 4) break;   /* more than 20 % narrower */
					f1 = std::max(f1, len1[x]);
					++b1;
				}
				rc = align_device(h, mode, b1 - b0, d_words, bits, d_woff1, d_len1, d_woff2, d_len2, f1, l2first, 0, tb ? 1 : 0,
				                  d_score, d_ei, d_ej, d_st, tb ? d_ops : nullptr, tb ? d_opsoff : nullptr, tb ? d_nops : nullptr, s,
				                  0, 0, d_order + b0, g);
				if (rc) return rc;
				b0 = b1;
				++nb;
			}
			snprintf(h->cfg + strlen(h->cfg), sizeof h->cfg - strlen(h->cfg), " (%d frames)", nb);
		} else if (frames) {
			int nb = 0;
			auto real = [&](size_t q) { return order[q] < 0 ? ~order[q] : order[q]; };   /* (a padding repeat is ~index) */
			for (size_t b0 = 0; b0 < order.size();) {   /* one launch per (group width, rows per lane) */
				const int g = gclass(len1[real(b0)]), kc = kclass(len1[real(b0)]);
				size_t b1 = b0;
				int f1 = 0, f2 = 0;
				while (b1 < order.size() && gclass(len1[real(b1)]) == g && kclass(len1[real(b1)]) == kc) {
					f1 = std::max(f1, len1[real(b1)]); f2 = std::max(f2, len2[real(b1)]);
					++b1;
				}
				rc = align_device(h, mode, (int64_t)(b1 - b0), d_words, bits, d_woff1, d_len1, d_woff2, d_len2, f1, f2, 0, tb ? 1 : 0,
				                  d_score, d_ei, d_ej, d_st, tb ? d_ops : nullptr, tb ? d_opsoff : nullptr, tb ? d_nops : nullptr, s,
				                  0, 0, d_order + b0, g);
				if (rc) return rc;
				b0 = b1;
				++nb;
			}
			snprintf(h->cfg + strlen(h->cfg), sizeof h->cfg - strlen(h->cfg), " (%d frames, equal-l1 work items)", nb);
		}
	}
	if (!frames) {
		rc = align_device(h, mode, npairs, d_words, bits, d_woff1, d_len1, d_woff2, d_len2, max1, max2, uniform ? 1 : 0, tb ? 1 : 0,
		                  d_score, d_ei, d_ej, d_st, tb ? d_ops : nullptr, tb ? d_opsoff : nullptr, tb ? d_nops : nullptr, s, 0, 0, d_order);
		if (rc) return rc;
	}
	htrace("chunk: sweep queued, pair", pair_base);
	/* ---- results: the five fixed-size arrays come down in ONE copy into page-locked memory.  Only the bytes of each pair's own
	 * ---- traceback travel and are written: the used part of every ops slot (or string slot) is packed back to back on the GPU and
	 * ---- comes down with the same synchronisation -- as many bytes as the previous call's payload suggests, the rest (if this
	 * ---- batch's tracebacks are longer) behind a second one -- and is scattered into the caller's slots here.  Bytes of the
	 * ---- caller's buffers between and behind the slots are never touched, whatever order the slots are in. ---- */
	const size_t res_bytes = 5 * b_len1;
	const size_t b_str = al((size_t)slots_total + (size_t)npairs + 64);
	const size_t pk_cap = tb ? (strings ? 2 : 1) * b_str : 0;
	rc = grow_pinned(h, &h->hp_out, &h->hp_out_bytes, res_bytes + 64 + (tb ? b_pfx : 0) + pk_cap);
	if (rc) return rc;
	char *ho = (char *)h->hp_out;
	int *p_rflag = (int *)(ho + res_bytes);
	int64_t *h_poff = (int64_t *)(ho + res_bytes + 64);
	char *h_pk1 = ho + res_bytes + 64 + (tb ? b_pfx : 0), *h_pk2 = h_pk1 + b_str;
	HIP_TRY(h, hipMemcpyAsync(ho, dout, res_bytes, hipMemcpyDeviceToHost, s));
	*p_rflag = 0;
	uint8_t *d_pk1 = nullptr, *d_pk2 = nullptr;
	size_t spec = 0;                                   /* payload bytes fetched before the total is known */
	if (tb) {
		rc = grow(h, &h->d_str, &h->str_bytes, (strings ? 2 : 1) * b_str + (strings ? b_pfx : 0));
		if (rc) return rc;
		d_pk1 = (uint8_t *)h->d_str; d_pk2 = d_pk1 + b_str;
		if (!strings) {
			rc = at_compact_ops_device(h, npairs, d_ops, d_opsoff, d_nops, d_pk1, slots_total, d_poff, s);
			if (rc) return rc;
		} else {
			rc = scan_nops_device(h, npairs, d_nops, d_poff, s);
			if (rc) return rc;
			int64_t *d_stroff = (int64_t *)(d_pk1 + 2 * b_str);   /* string k starts at poff[k] + k: one NUL behind each */
			hipLaunchKernelGGL(at::at_add_index, dim3((unsigned)std::min<int64_t>((npairs + 255) / 256, 8LL * h->ncu)), dim3(256), 0, s,
			                   (const long long *)d_poff, (long long *)d_stroff, (long long)npairs);
			HIP_TRY(h, hipMemsetAsync(h->d_rflag, 0, 4, s));
			rc = at_render_batch_device(h, npairs, d_words, bits, d_woff1, d_woff2, d_ei, d_ej, d_ops, d_opsoff, d_nops,
			                            d_pk1, d_pk2, d_stroff, 1, s);
			if (rc) return rc;
			HIP_TRY(h, hipMemcpyAsync(p_rflag, h->d_rflag, 4, hipMemcpyDeviceToHost, s));
		}
		HIP_TRY(h, hipMemcpyAsync(h_poff, d_poff, (size_t)(npairs + 1) * 8, hipMemcpyDeviceToHost, s));
		spec = std::min<size_t>((size_t)slots_total + (strings ? (size_t)npairs : 0),
		                        (size_t)((double)h->last_payload_per_pair * 1.25 * (double)npairs) + 65536);
		HIP_TRY(h, hipMemcpyAsync(h_pk1, d_pk1, spec, hipMemcpyDeviceToHost, s));
		if (strings) HIP_TRY(h, hipMemcpyAsync(h_pk2, d_pk2, spec, hipMemcpyDeviceToHost, s));
	}
	HIP_TRY(h, hipStreamSynchronize(s));
	htrace("chunk: results down, pair", pair_base);
	const int32_t *r_score = (const int32_t *)ho, *r_nops = (const int32_t *)(ho + 4 * b_len1);
	for (int64_t k = 0; k < npairs; ++k) {
		if (r_score[k] == INT32_MIN || (tb && r_nops[k] < 0))
			return fail(h, AT_ERR_DOMAIN, "pair %lld: input outside the domain on which the reference is defined", (long long)(pair_base + k));
	}
	memcpy(out_score, ho, (size_t)npairs * 4);
	if (out_end_i) memcpy(out_end_i, ho + b_len1, (size_t)npairs * 4);
	if (out_end_j) memcpy(out_end_j, ho + 2 * b_len1, (size_t)npairs * 4);
	if (out_state) memcpy(out_state, ho + 3 * b_len1, (size_t)npairs * 4);
	if (tb) {
		memcpy(out_nops, r_nops, (size_t)npairs * 4);
		const int64_t total = h_poff[(size_t)npairs];
		if (total < 0 || total > slots_total) return fail(h, AT_ERR_DOMAIN, "traceback lengths inconsistent with the slots");
		const size_t have = (size_t)total + (strings ? (size_t)npairs : 0);
		h->last_payload_per_pair = (double)have / (double)npairs;
		if (have > spec) {       /* longer tracebacks than the last batch's: the rest of the payload */
			HIP_TRY(h, hipMemcpyAsync(h_pk1 + spec, d_pk1 + spec, have - spec, hipMemcpyDeviceToHost, s));
			if (strings) HIP_TRY(h, hipMemcpyAsync(h_pk2 + spec, d_pk2 + spec, have - spec, hipMemcpyDeviceToHost, s));
			HIP_TRY(h, hipStreamSynchronize(s));
		}
		if (!strings) {
			for (int64_t k = 0; k < npairs; ++k)
				if (r_nops[k] > 0) memcpy(out_ops + ops_off[k], h_pk1 + h_poff[(size_t)k], (size_t)r_nops[k]);
		} else {
			for (int64_t k = 0; k < npairs; ++k) {
				const size_t src = (size_t)(h_poff[(size_t)k] + k), nb = (size_t)r_nops[k] + 1;   /* with the NUL */
				memcpy(out_r1 + ops_off[k], h_pk1 + src, nb);
				memcpy(out_r2 + ops_off[k], h_pk2 + src, nb);
			}
		}
	}
	if (*p_rflag) return fail(h, AT_ERR_DOMAIN, "traceback inconsistent with its sequences");
	htrace("chunk: done, pair", pair_base);
	return AT_OK;
}

/* A large batch is cut into contiguous chunks that run side by side, each on its own handle (stream, workspace) and
 * host thread: the upload of one chunk, the sweep of another and the download of a third overlap, and the launches fill
 * each other's tails.  Chunks address the caller's buffers with the caller's absolute offsets. */
static int align_host_mt(at_handle *h, int mode, int64_t npairs, const uint8_t *seq_blob,
                         const int64_t *off1, const int32_t *len1, const int64_t *off2, const int32_t *len2,
                         int want_traceback,
                         int32_t *out_score, int32_t *out_end_i, int32_t *out_end_j, int32_t *out_state,
                         uint8_t *out_ops, const int64_t *ops_off, int32_t *out_nops, char *out_r1, char *out_r2)
{
	/* uniform batches: six chunks (100k pairs of C2: 3.37 ms in one piece, 2.49 in 3 chunks, 2.28 in 6); ragged ones, whose chunks are
	 * sorted into frames one by one: three (fit -s 100..150 x 400..500: 785 GCUPS in 3 chunks, 693 in 6 -- smaller frames) */
	htrace("batch: enter, pairs", npairs);
	bool same = len1 && len2;
	for (int64_t k = 1; same && k < npairs; ++k) same = len1[k] == len1[0] && len2[k] == len2[0];
	const long long want = env_ll("AT_HOST_CHUNKS", same ? 6 : 3);
	const long long min_pairs = env_ll("AT_HOST_CHUNK_MIN", 16384);
	int nchunks = (int)std::max<long long>(1, std::min<long long>(want, 12));
	if (!h || npairs < 2 * min_pairs || !seq_blob || !off1 || !len1 || !off2 || !len2 || !out_score) nchunks = 1;
	else nchunks = (int)std::min<long long>(nchunks, npairs / min_pairs);
	if (nchunks <= 1)
		return align_host(h, mode, npairs, seq_blob, off1, len1, off2, len2, want_traceback, out_score, out_end_i, out_end_j,
		                  out_state, out_ops, ops_off, out_nops, out_r1, out_r2);
	while ((int)h->kids.size() < nchunks - 1) {
		at_handle *k = nullptr;
		const int dev = h->device;
		const int rc = at_init(&dev, 1, &k);
		if (rc) return fail(h, rc, "helper handle: %s", at_last_error(nullptr));
		h->kids.push_back(k);
	}
	for (int c = 0; c < nchunks - 1; ++c) {
		at_handle *k = h->kids[(size_t)c];
		if (k->m != h->m || k->u != h->u || k->o != h->o || k->e != h->e || k->j != h->j || k->use_jump != h->use_jump ||
		    k->sites != h->sites) {
			k->m = h->m; k->u = h->u; k->o = h->o; k->e = h->e; k->j = h->j; k->use_jump = h->use_jump; k->sites = h->sites;
			k->sitemask_dirty = true;
		}
	}
	std::vector<int> rcs((size_t)nchunks, AT_OK);
	/* (Tried: unequal chunks -- small first ones so that the GPU starts early, small last ones so that the tail behind the last upload is
	 * short: 1.95 .. 2.21 ms against 2.02 for equal shares; the box's noise.) */
	const int64_t per = ((npairs + nchunks - 1) / nchunks + 7) & ~(int64_t)7;
	auto run = [&](int c) {
		const int64_t lo = std::min<int64_t>(npairs, c * per), n = std::min<int64_t>(npairs, lo + per) - lo;
		at_handle *hh = c == 0 ? h : h->kids[(size_t)c - 1];
		if (n <= 0) return;
		/* (a chunk runs on a pool thread: an exception that left it would end the process in std::terminate) */
		rcs[(size_t)c] = guarded(hh, "at_align_batch", [&] {
			return align_host(hh, mode, n, seq_blob, off1 + lo, len1 + lo, off2 + lo, len2 + lo, want_traceback,
			                  out_score + lo, out_end_i ? out_end_i + lo : nullptr, out_end_j ? out_end_j + lo : nullptr,
			                  out_state ? out_state + lo : nullptr, out_ops, ops_off ? ops_off + lo : nullptr,
			                  out_nops ? out_nops + lo : nullptr, out_r1, out_r2, lo);
		});
	};
	if (!h->pool) h->pool = new HostPool();
	h->pool->run(nchunks, run);
	for (int c = 0; c < nchunks; ++c) {
		if (rcs[(size_t)c] != AT_OK) {
			if (c > 0) snprintf(h->err, sizeof h->err, "%s", h->kids[(size_t)c - 1]->err);
			snprintf(g_err, sizeof g_err, "%s", h->err);
			return rcs[(size_t)c];
		}
	}
	snprintf(h->cfg + strlen(h->cfg), sizeof h->cfg - strlen(h->cfg), " x%d chunks", nchunks);
	return AT_OK;
}

/* ---- all-vs-all over one read set held in HOST memory ----
 * The reads go up once and are packed once (upload_reads); pairs of the strict upper triangle are enumerated on the GPU
 * (at_align_allpairs_device).  The triangle is swept in SLICES of at most `chunk` pairs, so device and host memory are
 * bounded by the slice, not by the 1.25e9 pairs of C5: scores and end cells of slice c come down on a copy stream and are
 * handed to the caller while the sweep of slice c + 1 runs. */
struct ReadSet {
	uint32_t *d_words = nullptr;
	int64_t *d_swoff = nullptr;
	int32_t *d_len = nullptr;
	int bits = 2, maxlen = 0;
};

static int upload_reads(at_handle *h, int mode, int64_t nreads, const uint8_t *seq_blob, const int64_t *off, const int32_t *len, ReadSet *rs)
{
	int maxlen = 0;
	int64_t blob_lo = INT64_MAX, blob_bytes = 0, nwords2 = 0, nwords8 = 0;
	std::vector<int64_t> soff((size_t)nreads), swoff2((size_t)nreads), swoff8((size_t)nreads);
	for (int64_t k = 0; k < nreads; ++k) {
		if (len[k] < 0 || off[k] < 0) return fail(h, AT_ERR_ARG, "read %lld: negative length or offset", (long long)k);
		if ((mode == AT_MODE_LOCAL || mode == AT_MODE_OVERLAP) && len[k] < 1) return fail(h, AT_ERR_DOMAIN, "read %lld: empty", (long long)k);
		maxlen = std::max(maxlen, len[k]);
		blob_lo = std::min(blob_lo, off[k]);
	}
	for (int64_t k = 0; k < nreads; ++k) {
		soff[(size_t)k] = off[k] - blob_lo;
		swoff2[(size_t)k] = nwords2; nwords2 += (len[k] + 15) / 16 + 1;
		swoff8[(size_t)k] = nwords8; nwords8 += (len[k] + 3) / 4 + 1;
		blob_bytes = std::max<int64_t>(blob_bytes, soff[(size_t)k] + len[k]);
	}
	HIP_TRY(h, hipSetDevice(h->device));
	auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t b_words = al((size_t)(std::max(nwords2, nwords8) + 4) * 4), b_off = al((size_t)nreads * 8), b_len = al((size_t)nreads * 4);
	const size_t b_blob = al((size_t)blob_bytes + 32);   /* (at_pack reads whole dwords: up to 20 bytes behind the last base) */
	int rc = grow(h, &h->d_in, &h->in_bytes, b_words + 2 * b_off + b_len + 256 + b_blob);
	if (rc) return rc;
	char *din = (char *)h->d_in;
	uint32_t *d_words = (uint32_t *)din;
	int64_t *d_soff = (int64_t *)(din + b_words), *d_swoff = (int64_t *)(din + b_words + b_off);
	int32_t *d_len = (int32_t *)(din + b_words + 2 * b_off);
	int *d_flag = (int *)(din + b_words + 2 * b_off + b_len);
	uint8_t *d_blob = (uint8_t *)(din + b_words + 2 * b_off + b_len + 256);
	hipStream_t s = h->stream;
	HIP_TRY(h, hipMemcpyAsync(d_blob, seq_blob + blob_lo, (size_t)blob_bytes, hipMemcpyHostToDevice, s));
	HIP_TRY(h, hipMemcpyAsync(d_soff, soff.data(), (size_t)nreads * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(h, hipMemcpyAsync(d_len, len, (size_t)nreads * 4, hipMemcpyHostToDevice, s));
	at::PackArgs pa;
	pa.nseq = nreads; pa.blob = d_blob; pa.off = (const long long *)d_soff; pa.len = d_len;
	pa.woff = (const long long *)d_swoff; pa.words = d_words; pa.not_acgt = d_flag;
	const unsigned pgrid = (unsigned)std::min<int64_t>((nreads + 15) / 16, 8LL * h->ncu);
	int bits = scores_fit_byte(h, mode) ? 2 : 8, flag = 0;
	if (bits == 2) {
		HIP_TRY(h, hipMemsetAsync(d_flag, 0, 4, s));
		HIP_TRY(h, hipMemcpyAsync(d_swoff, swoff2.data(), (size_t)nreads * 8, hipMemcpyHostToDevice, s));
		hipLaunchKernelGGL(at::at_pack<2>, dim3(pgrid), dim3(256), 0, s, pa);
		HIP_TRY(h, hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(h, hipStreamSynchronize(s));
		if (flag) bits = 8;
	}
	if (bits == 8) {
		HIP_TRY(h, hipMemcpyAsync(d_swoff, swoff8.data(), (size_t)nreads * 8, hipMemcpyHostToDevice, s));
		hipLaunchKernelGGL(at::at_pack<8>, dim3(pgrid), dim3(256), 0, s, pa);
	}
	HIP_TRY(h, hipGetLastError());
	HIP_TRY(h, hipStreamSynchronize(s));   /* swoff* are stack-lifetime vectors */
	rs->d_words = d_words; rs->d_swoff = d_swoff; rs->d_len = d_len; rs->bits = bits; rs->maxlen = maxlen;
	return AT_OK;
}

static int check_allpairs_args(at_handle *h, const char *who, int mode, int64_t nreads, int64_t first_pair, int64_t npairs)
{
	if (!h) return fail(nullptr, AT_ERR_ARG, "%s: NULL handle", who);
	if (mode < AT_MODE_GLOBAL || mode > AT_MODE_EDIT) return fail(h, AT_ERR_ARG, "unknown mode %d", mode);
	if (mode == AT_MODE_FIT) return fail(h, AT_ERR_ARG, "all-vs-all: fit needs ordered lengths (l1 <= l2); use the pair list entry");
	if (nreads < 2 || nreads >= (1LL << 31) || first_pair < 0 || npairs < 0 || first_pair + npairs > nreads * (nreads - 1) / 2)
		return fail(h, AT_ERR_ARG, "all-vs-all: pair range [%lld, +%lld) outside the %lld*(%lld-1)/2 ordered pairs",
		            (long long)first_pair, (long long)npairs, (long long)nreads, (long long)nreads);
	return AT_OK;
}

/* scores + end cells of pairs [first_pair, first_pair + npairs) in slices; fn(user, first pair of the slice, pairs, score, end_i,
 * end_j, state) sees every slice once, in order, on the calling thread */
static int allpairs_scores(at_handle *h, int mode, int64_t nreads, const ReadSet &rs, int64_t first_pair, int64_t npairs,
                           int64_t chunk, at_allpairs_chunk_fn fn, void *user)
{
	if (chunk <= 0) chunk = env_ll("AT_ALLPAIRS_CHUNK", 4LL << 20);
	chunk = std::max<int64_t>(1, std::min<int64_t>(chunk, std::min<int64_t>(npairs, 1LL << 28)));
	auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t b_res = al((size_t)chunk * 4);
	int rc = grow(h, &h->d_out, &h->out_bytes, 2 * 4 * b_res);
	if (rc) return rc;
	if (h->pin_bytes < 2 * 4 * b_res) {
		if (h->h_pin) { (void)hipHostFree(h->h_pin); h->h_pin = nullptr; h->pin_bytes = 0; }
		if (hipHostMalloc(&h->h_pin, 2 * 4 * b_res, hipHostMallocDefault) != hipSuccess)
			return fail(h, AT_ERR_NOMEM, "hipHostMalloc(%zu) for the result slices failed", 2 * 4 * b_res);
		h->pin_bytes = 2 * 4 * b_res;
	}
	if (!h->copy_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
	for (int q = 0; q < 2; ++q) {
		if (!h->ev_sweep[q]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_sweep[q], hipEventDisableTiming));
		if (!h->ev_copy[q]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_copy[q], hipEventDisableTiming));
	}
	hipStream_t s = h->stream, cs = h->copy_stream;
	const int64_t nchunks = (npairs + chunk - 1) / chunk;
	auto dres = [&](int q, int a) { return (int32_t *)((char *)h->d_out + ((size_t)q * 4 + (size_t)a) * b_res); };
	auto hres = [&](int q, int a) { return (int32_t *)((char *)h->h_pin + ((size_t)q * 4 + (size_t)a) * b_res); };
	auto deliver = [&](int64_t c) -> int {
		const int q = (int)(c & 1);
		const int64_t lo = c * chunk, n = std::min(chunk, npairs - lo);
		HIP_TRY(h, hipEventSynchronize(h->ev_copy[q]));
		const int32_t *sc = hres(q, 0);
		for (int64_t p = 0; p < n; ++p)
			if (sc[p] == INT32_MIN)
				return fail(h, AT_ERR_DOMAIN, "pair %lld: input outside the domain on which the reference is defined", (long long)(first_pair + lo + p));
		if (fn && fn(user, first_pair + lo, n, sc, hres(q, 1), hres(q, 2), hres(q, 3)) != 0)
			return fail(h, AT_ERR_ARG, "all-vs-all: the caller's slice callback asked to stop at pair %lld", (long long)(first_pair + lo));
		return AT_OK;
	};
	std::string cfg0;
	for (int64_t c = 0; c < nchunks; ++c) {
		const int q = (int)(c & 1);
		const int64_t lo = c * chunk, n = std::min(chunk, npairs - lo);
		/* (slice c - 2, the last user of buffer set q, was delivered before slice c - 1 was queued) */
		rc = align_device(h, mode, n, rs.d_words, rs.bits, rs.d_swoff, rs.d_len, rs.d_swoff, rs.d_len, rs.maxlen, rs.maxlen, 0, 0,
		                  dres(q, 0), dres(q, 1), dres(q, 2), dres(q, 3), nullptr, nullptr, nullptr, s, nreads, first_pair + lo);
		if (rc) { (void)hipDeviceSynchronize(); return rc; }
		if (c == 0) cfg0 = h->cfg;
		HIP_TRY(h, hipEventRecord(h->ev_sweep[q], s));
		HIP_TRY(h, hipStreamWaitEvent(cs, h->ev_sweep[q], 0));
		for (int a = 0; a < 4; ++a) HIP_TRY(h, hipMemcpyAsync(hres(q, a), dres(q, a), (size_t)n * 4, hipMemcpyDeviceToHost, cs));
		HIP_TRY(h, hipEventRecord(h->ev_copy[q], cs));
		if (c >= 1) {
			rc = deliver(c - 1);
			if (rc) { (void)hipDeviceSynchronize(); return rc; }
		}
	}
	rc = deliver(nchunks - 1);
	if (rc) return rc;
	snprintf(h->cfg, sizeof h->cfg, "%.260s, %lld slices of <= %lld pairs", cfg0.c_str(), (long long)nchunks, (long long)chunk);
	return AT_OK;
}

extern "C" int at_align_allpairs_stream(at_handle *h, int mode, int64_t nreads, const uint8_t *seq_blob,
                                        const int64_t *off, const int32_t *len, int64_t first_pair, int64_t npairs,
                                        int64_t chunk_pairs, at_allpairs_chunk_fn fn, void *user)
{
	int rc = check_allpairs_args(h, "at_align_allpairs_stream", mode, nreads, first_pair, npairs);
	if (rc) return rc;
	if (npairs == 0) return AT_OK;
	if (!seq_blob || !off || !len || !fn) return fail(h, AT_ERR_ARG, "NULL argument");
	return guarded(h, "at_align_allpairs_stream", [&] {
		ReadSet rs;
		int rc2 = upload_reads(h, mode, nreads, seq_blob, off, len, &rs);
		if (rc2) return rc2;
		return allpairs_scores(h, mode, nreads, rs, first_pair, npairs, chunk_pairs, fn, user);
	});
}

namespace {
struct CopyOut { int64_t first; int32_t *score, *end_i, *end_j, *state; };
int copy_out_slice(void *user, int64_t first, int64_t n, const int32_t *sc, const int32_t *ei, const int32_t *ej, const int32_t *st)
{
	const CopyOut *o = (const CopyOut *)user;
	const size_t at = (size_t)(first - o->first), nb = (size_t)n * 4;
	memcpy(o->score + at, sc, nb);
	if (o->end_i) memcpy(o->end_i + at, ei, nb);
	if (o->end_j) memcpy(o->end_j + at, ej, nb);
	if (o->state) memcpy(o->state + at, st, nb);
	return 0;
}
}

/* one slice of the triangle WITH tracebacks: ops of pair p go to the caller's slot ops_off[p - first_pair] */
static int allpairs_tb_slice(at_handle *h, int mode, int64_t nreads, const ReadSet &rs, int64_t first_pair, int64_t npairs,
                             int32_t *out_score, int32_t *out_end_i, int32_t *out_end_j, int32_t *out_state,
                             uint8_t *out_ops, const int64_t *ops_off, int32_t *out_nops)
{
	const int maxlen = rs.maxlen;
	int64_t ops_total = 0, ops_lo = INT64_MAX;
	for (int64_t p = 0; p < npairs; ++p) { if (ops_off[p] < 0) return fail(h, AT_ERR_ARG, "pair %lld: negative ops offset", (long long)(first_pair + p)); ops_lo = std::min(ops_lo, ops_off[p]); }
	std::vector<int64_t> opsr((size_t)npairs);
	for (int64_t p = 0; p < npairs; ++p) { opsr[(size_t)p] = ops_off[p] - ops_lo; ops_total = std::max<int64_t>(ops_total, opsr[(size_t)p] + 2LL * maxlen); }
	const int64_t slots_total = npairs * 2LL * maxlen;
	auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
	int rc = grow(h, &h->d_desc, &h->desc_bytes, al((size_t)npairs * 8));
	if (rc) return rc;
	int64_t *d_opsoff = (int64_t *)h->d_desc;
	hipStream_t s = h->stream;
	HIP_TRY(h, hipMemcpyAsync(d_opsoff, opsr.data(), (size_t)npairs * 8, hipMemcpyHostToDevice, s));
	const size_t b_res = al((size_t)npairs * 4), b_ops = al((size_t)ops_total + 64), b_pfx = al((size_t)(npairs + 1) * 8);
	rc = grow(h, &h->d_out, &h->out_bytes, 5 * b_res + b_ops + b_pfx);
	if (rc) return rc;
	char *dout = (char *)h->d_out;
	int32_t *d_score = (int32_t *)dout, *d_ei = (int32_t *)(dout + b_res), *d_ej = (int32_t *)(dout + 2 * b_res);
	int32_t *d_st = (int32_t *)(dout + 3 * b_res), *d_nops = (int32_t *)(dout + 4 * b_res);
	uint8_t *d_ops = (uint8_t *)(dout + 5 * b_res);
	int64_t *d_poff = (int64_t *)(dout + 5 * b_res + b_ops);
	rc = align_device(h, mode, npairs, rs.d_words, rs.bits, rs.d_swoff, rs.d_len, rs.d_swoff, rs.d_len, maxlen, maxlen, 0, 1, d_score, d_ei, d_ej, d_st,
	                  d_ops, d_opsoff, d_nops, s, nreads, first_pair);
	if (rc) return rc;
	HIP_TRY(h, hipMemcpyAsync(out_score, d_score, (size_t)npairs * 4, hipMemcpyDeviceToHost, s));
	if (out_end_i) HIP_TRY(h, hipMemcpyAsync(out_end_i, d_ei, (size_t)npairs * 4, hipMemcpyDeviceToHost, s));
	if (out_end_j) HIP_TRY(h, hipMemcpyAsync(out_end_j, d_ej, (size_t)npairs * 4, hipMemcpyDeviceToHost, s));
	if (out_state) HIP_TRY(h, hipMemcpyAsync(out_state, d_st, (size_t)npairs * 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(h, hipMemcpyAsync(out_nops, d_nops, (size_t)npairs * 4, hipMemcpyDeviceToHost, s));
	rc = grow(h, &h->d_str, &h->str_bytes, al((size_t)slots_total + 64));
	if (rc) return rc;
	rc = at_compact_ops_device(h, npairs, d_ops, d_opsoff, d_nops, (uint8_t *)h->d_str, slots_total, d_poff, s);
	if (rc) return rc;
	std::vector<int64_t> h_poff((size_t)npairs + 1);
	HIP_TRY(h, hipMemcpyAsync(h_poff.data(), d_poff, (size_t)(npairs + 1) * 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(h, hipStreamSynchronize(s));
	for (int64_t p = 0; p < npairs; ++p)
		if (out_score[p] == INT32_MIN || out_nops[p] < 0)
			return fail(h, AT_ERR_DOMAIN, "pair %lld: input outside the domain on which the reference is defined", (long long)(first_pair + p));
	const int64_t total = h_poff[(size_t)npairs];
	if (total < 0 || total > slots_total) return fail(h, AT_ERR_DOMAIN, "traceback lengths inconsistent with the slots");
	std::vector<uint8_t> pk((size_t)total + 1);
	if (total) HIP_TRY(h, hipMemcpyAsync(pk.data(), h->d_str, (size_t)total, hipMemcpyDeviceToHost, s));
	HIP_TRY(h, hipStreamSynchronize(s));
	for (int64_t p = 0; p < npairs; ++p)
		if (out_nops[p] > 0) memcpy(out_ops + ops_off[p], pk.data() + h_poff[(size_t)p], (size_t)out_nops[p]);
	return AT_OK;
}

extern "C" int at_align_allpairs(at_handle *h, int mode, int64_t nreads, const uint8_t *seq_blob,
                                 const int64_t *off, const int32_t *len, int64_t first_pair, int64_t npairs, int want_traceback,
                                 int32_t *out_score, int32_t *out_end_i, int32_t *out_end_j, int32_t *out_state,
                                 uint8_t *out_ops, const int64_t *ops_off, int32_t *out_nops)
{
	int rc = check_allpairs_args(h, "at_align_allpairs", mode, nreads, first_pair, npairs);
	if (rc) return rc;
	if (npairs == 0) return AT_OK;
	if (!seq_blob || !off || !len || !out_score) return fail(h, AT_ERR_ARG, "NULL argument");
	const bool tb = want_traceback && mode != AT_MODE_EDIT;
	if (tb && (!out_ops || !ops_off || !out_nops)) return fail(h, AT_ERR_ARG, "traceback wanted but ops buffers are NULL");
	return guarded(h, "at_align_allpairs", [&] {
		ReadSet rs;
		int rc2 = upload_reads(h, mode, nreads, seq_blob, off, len, &rs);
		if (rc2) return rc2;
		if (!tb) {
			CopyOut co = {first_pair, out_score, out_end_i, out_end_j, out_state};
			return allpairs_scores(h, mode, nreads, rs, first_pair, npairs, 0, copy_out_slice, &co);
		}
		/* with tracebacks: slices whose ops slots (2 * maxlen bytes per pair) stay below ~1 GiB of device memory */
		const int64_t per = std::max<int64_t>(1, std::min<int64_t>(1LL << 22, (1LL << 30) / std::max(1, 2 * rs.maxlen)));
		for (int64_t lo = 0; lo < npairs; lo += per) {
			const int64_t n = std::min(per, npairs - lo);
			rc2 = allpairs_tb_slice(h, mode, nreads, rs, first_pair + lo, n, out_score + lo, out_end_i ? out_end_i + lo : nullptr,
			                        out_end_j ? out_end_j + lo : nullptr, out_state ? out_state + lo : nullptr, out_ops, ops_off + lo, out_nops + lo);
			if (rc2) return rc2;
		}
		return (int)AT_OK;
	});
}

extern "C" int at_align_batch(at_handle *h, int mode, int64_t npairs, const uint8_t *seq_blob,
                              const int64_t *off1, const int32_t *len1, const int64_t *off2, const int32_t *len2,
                              int want_traceback,
                              int32_t *out_score, int32_t *out_end_i, int32_t *out_end_j, int32_t *out_state,
                              uint8_t *out_ops, const int64_t *ops_off, int32_t *out_nops)
{
	return guarded(h, "at_align_batch", [&] {
		return align_host_mt(h, mode, npairs, seq_blob, off1, len1, off2, len2, want_traceback, out_score, out_end_i, out_end_j,
		                     out_state, out_ops, ops_off, out_nops, nullptr, nullptr);
	});
}

extern "C" int at_align_batch_strings(at_handle *h, int mode, int64_t npairs, const uint8_t *seq_blob,
                                      const int64_t *off1, const int32_t *len1, const int64_t *off2, const int32_t *len2,
                                      int32_t *out_score, int32_t *out_end_i, int32_t *out_end_j, int32_t *out_state,
                                      char *out_r1, char *out_r2, const int64_t *str_off, int32_t *out_len)
{
	if (mode == AT_MODE_EDIT) return fail(h, AT_ERR_ARG, "edit has no alignment strings (alignment.h:291)");
	if (!out_r1 || !out_r2 || !str_off || !out_len) return fail(h, AT_ERR_ARG, "NULL string buffers");
	return guarded(h, "at_align_batch_strings", [&] {
		return align_host_mt(h, mode, npairs, seq_blob, off1, len1, off2, len2, 1, out_score, out_end_i, out_end_j, out_state,
		                     nullptr, str_off, out_len, out_r1, out_r2);
	});
}
