/*
 * ref_harness.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Thin driver around the REAL reference implementation: it #include's
 * /root/reference/src/alignment.h where it lies (nothing is copied into this
 * repo) and exposes the five reference kernels
 *     align_gla               alignment.h:417
 *     align_local_affine      alignment.h:805
 *     align_fit_affine_jump   alignment.h:596
 *     align_overlap           alignment.h:926
 *     edit_dist               alignment.h:291
 * through a flat C ABI that python/ctypes can call.  It is built by
 * oracle/Makefile into oracle/_ref/libat_ref.so (git-ignored, shipped to the
 * GPU box as a binary) and is used to
 *   (1) generate tests/golden fixtures (oracle/make_golden.py),
 *   (2) validate the C restatement in oracle/at_oracle.c,
 *   (3) serve as bench.py's cpu_baseline (kind "reference").
 *
 * Two harness-level macros change NO arithmetic:
 *   - calloc is padded by 16 elements: strrev (alignment.h:178-182) allocates
 *     l bytes and writes s[l]; the stock binary corrupts its heap whenever the
 *     output length is 8 mod 16.
 *   - printf is swallowed: align_fit_affine_jump prints a debug line to stdout
 *     (alignment.h:602).
 * <stdbool.h> must not be included: alignment.h:24 defines its own
 * `enum { true, false }` (true == 0).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static int ref_printf_sink(const char *fmt, ...) { (void)fmt; return 0; }
#define printf ref_printf_sink
#define calloc(a, b) calloc((size_t)(a) + 16, (b))
#include "alignment.h"
#undef calloc
#undef printf

enum { REF_GLOBAL = 0, REF_LOCAL = 1, REF_FIT = 2, REF_OVERLAP = 3, REF_EDIT = 4 };

static kstring_t *mk_ks(const char *s, int l)
{
	kstring_t *k = (kstring_t *)calloc(1, sizeof(kstring_t));
	k->s = (char *)calloc((size_t)l + 17, 1);
	memcpy(k->s, s, (size_t)l);
	k->l = (size_t)l;
	k->m = (size_t)l + 17;
	return k;
}

static kstring_t *mk_res(int cap)
{
	kstring_t *k = (kstring_t *)calloc(1, sizeof(kstring_t));
	k->s = (char *)calloc((size_t)cap + 17, 1);
	return k;
}

/*
 * One pair through the reference.  Returns 0, or -1 for bad arguments.
 * r1/r2 (capacity cap >= l1+l2+1) receive the two gapped strings, NUL
 * terminated; *rlen their length.  For REF_EDIT the distance is returned in
 * *score and r1/r2 are left empty.
 */
int ref_align(int mode, const char *s1, int l1, const char *s2, int l2,
              int m, int u, int o, int e, int j, int use_jump,
              const int *sites, int nsites,
              double *score, char *r1, char *r2, int cap, int *rlen)
{
	if (l1 < 0 || l2 < 0 || cap < l1 + l2 + 1) return -1;
	if (mode == REF_FIT && l1 > l2) return -1; /* reference die()s: alignment.h:599 */
	opt_t *opt = init_opt();
	opt->m = m; opt->u = u; opt->o = o; opt->e = e; opt->j = j;
	opt->s = use_jump ? true : false; /* NB: `true` is 0 in the reference's enum */
	opt->sites.size = (size_t)nsites;
	opt->sites.pos = (int *)sites;
	kstring_t *k1 = mk_ks(s1, l1), *k2 = mk_ks(s2, l2);
	r1[0] = r2[0] = 0;
	*rlen = 0;
	if (mode == REF_EDIT) {
		*score = (double)edit_dist(k1, k2, opt);
	} else {
		kstring_t *a = mk_res(l1 + l2), *b = mk_res(l1 + l2);
		switch (mode) {
		case REF_GLOBAL:  *score = align_gla(k1, k2, a, b, opt); break;
		case REF_LOCAL:   *score = align_local_affine(k1, k2, a, b, opt); break;
		case REF_FIT:     *score = align_fit_affine_jump(k1, k2, a, b, opt); break;
		case REF_OVERLAP: *score = align_overlap(k1, k2, a, b, opt); break;
		default: return -1;
		}
		*rlen = (int)a->l;
		memcpy(r1, a->s, a->l); r1[a->l] = 0;
		memcpy(r2, b->s, b->l); r2[b->l] = 0;
		kstring_destory(a);
		kstring_destory(b);
	}
	kstring_destory(k1);
	kstring_destory(k2);
	free(opt);
	return 0;
}

/*
 * Timed loop for the CPU baseline: pairs [0,n) of fixed shape l1 x l2 laid
 * out back to back in `blob` (s1 then s2 per pair).  Returns wall seconds
 * (CLOCK_MONOTONIC) and a checksum of scores so the work cannot be elided.
 */
double ref_time_batch(int mode, int n, const char *blob, int l1, int l2,
                      int m, int u, int o, int e, int j, int use_jump,
                      const int *sites, int nsites, double *checksum)
{
	struct timespec t0, t1;
	int cap = l1 + l2 + 1, rl;
	char *r1 = (char *)malloc((size_t)cap), *r2 = (char *)malloc((size_t)cap);
	double acc = 0, sc;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (int k = 0; k < n; ++k) {
		const char *p = blob + (size_t)k * (size_t)(l1 + l2);
		ref_align(mode, p, l1, p + l1, l2, m, u, o, e, j, use_jump, sites, nsites,
		          &sc, r1, r2, cap, &rl);
		acc += sc + rl;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(r1); free(r2);
	*checksum = acc;
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
