/*
 * at_oracle.c -- TEST INFRASTRUCTURE ONLY (the parity checker; see at_oracle.h).
 *
 * A from-the-spec CPU restatement of the reference's DP fill + traceback.
 * It follows the reference's arithmetic exactly -- fp64 cells holding
 * integers or -inf, row-major i-outer/j-inner fill, first-wins strict-'>'
 * arg-max, pointer matrices, pointer-chasing traceback -- but is written
 * independently (flat matrices, one generic first-wins helper, op codes
 * instead of in-place string building).  Each routine cites the reference
 * lines it restates.  "parity pinned": see at_oracle.h.
 */
#define _POSIX_C_SOURCE 200809L
#include "at_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define NEG (-INFINITY)

/* pointer-cell states (the reference uses LOW/MID/UPP/HOME/JUMP = 500..800,
 * alignment.h:27-34; 0 = never written = calloc'ed zero in the reference) */
enum { P_NONE = 0, P_LOW = 1, P_MID = 2, P_UPP = 3, P_HOME = 4, P_JUMP = 5,
       P_LEFT = 6, P_DIAG = 7, P_RIGHT = 8 };

/* max5 (alignment.h:90-100): running max starts at -inf, a candidate wins only
 * if strictly greater, so the FIRST of equal candidates wins and -inf never
 * wins.  Returns the winning index or -1 (the reference leaves `state`
 * uninitialised in that case). */
static int first_max(double *res, const double *c, int n)
{
	int k, st = -1;
	*res = NEG;
	for (k = 0; k < n; ++k)
		if (c[k] > *res) { *res = c[k]; st = k; }
	return st;
}

typedef struct {
	int rows, cols;          /* l1+1, l2+1 */
	double *L, *M, *U, *J;
	unsigned char *pL, *pM, *pU, *pJ;
} mats;

#define AT(mat, i, j) (mat)[(size_t)(i) * (size_t)W.cols + (size_t)(j)]

/* create_matrix (alignment.h:119-148): everything zero-initialised */
static int mats_new(mats *w, int l1, int l2, int with_j)
{
	size_t n = (size_t)(l1 + 1) * (size_t)(l2 + 1);
	memset(w, 0, sizeof *w);
	w->rows = l1 + 1; w->cols = l2 + 1;
	w->L = calloc(n, sizeof(double)); w->M = calloc(n, sizeof(double));
	w->U = calloc(n, sizeof(double));
	w->pL = calloc(n, 1); w->pM = calloc(n, 1); w->pU = calloc(n, 1);
	if (with_j) { w->J = calloc(n, sizeof(double)); w->pJ = calloc(n, 1); }
	if (!w->L || !w->M || !w->U || !w->pL || !w->pM || !w->pU) return -1;
	if (with_j && (!w->J || !w->pJ)) return -1;
	return 0;
}

static void mats_free(mats *w)
{
	free(w->L); free(w->M); free(w->U); free(w->J);
	free(w->pL); free(w->pM); free(w->pU); free(w->pJ);
}

typedef struct {
	unsigned char *ops;
	int n, cap;
} opbuf;

static int emit(opbuf *b, int op)
{
	if (b->n >= b->cap) return -1;
	b->ops[b->n++] = (unsigned char)op;
	return 0;
}

/* Build the two gapped strings from ops (traceback order) -- what the
 * reference does char by char followed by strrev (alignment.h:172-184). */
static void render(const unsigned char *ops, int n, const char *s1, const char *s2,
                   int i, int j, char *r1, char *r2)
{
	int k;
	for (k = 0; k < n; ++k) {
		int pos = n - 1 - k;
		switch (ops[k]) {
		case ATO_OP_MID: r1[pos] = s1[--i]; r2[pos] = s2[--j]; break;
		case ATO_OP_LOW: r1[pos] = s1[--i]; r2[pos] = '-'; break;
		default:         r1[pos] = '-';     r2[pos] = s2[--j]; break;
		}
	}
	r1[n] = r2[n] = 0;
}

/* shared 3-state (+J) pointer walk: trace_back_gla :372-412, _local_affine
 * :766-800, _fit_affine_jump :558-592.  `stop_on_j`: loop condition is
 * (i>0 && j>0) for global/local, (i>0) for fit. */
static int walk_affine(const mats *w, int i, int j, int state, int need_j_pos, opbuf *b,
                       int *oi, int *oj)
{
	const mats W = *w;
	long guard = 4L * (W.rows + W.cols) + 16;
	while (i > 0 && (!need_j_pos || j > 0)) {
		if (--guard < 0) return -2;
		if (j < 0) return -2;
		switch (state) {
		case P_LOW:  state = AT(W.pL, i, j); if (emit(b, ATO_OP_LOW)) return -2; --i; break;
		case P_MID:  state = AT(W.pM, i, j); if (emit(b, ATO_OP_MID)) return -2; --i; --j; break;
		case P_UPP:  state = AT(W.pU, i, j); if (emit(b, ATO_OP_UPP)) return -2; --j; break;
		case P_JUMP: state = AT(W.pJ, i, j); if (emit(b, ATO_OP_JUMP)) return -2; --j; break;
		case P_HOME: i = 0; j = 0; break;   /* local only, :788-791 */
		default: return -2;                  /* reference would spin forever */
		}
	}
	*oi = i; *oj = j;
	return 0;
}

/* ---- global: align_gla alignment.h:417-473 -------------------------------- */
static int do_global(const char *s1, int l1, const char *s2, int l2, const ato_scoring *sc,
                     double *score, int *ei, int *ej, int *est, opbuf *b)
{
	mats W; int i, j, st, rc;
	double m = sc->m, u = sc->u, o = sc->o, e = sc->e, c[3];
	if (mats_new(&W, l1, l2, 0)) { mats_free(&W); return -1; }
	/* borders :428-441 */
	AT(W.M, 0, 0) = 0.0; AT(W.L, 0, 0) = o; AT(W.U, 0, 0) = o;
	for (i = 1; i <= l1; ++i) { AT(W.L, i, 0) = o + e * i; AT(W.M, i, 0) = NEG; AT(W.U, i, 0) = NEG; }
	for (j = 1; j <= l2; ++j) { AT(W.L, 0, j) = NEG; AT(W.M, 0, j) = NEG; AT(W.U, 0, j) = o + e * j; }
	/* fill :446-464 */
	for (i = 1; i <= l1; ++i)
		for (j = 1; j <= l2; ++j) {
			double s = (s1[i - 1] == s2[j - 1]) ? m : u;
			c[0] = AT(W.L, i - 1, j - 1) + s; c[1] = AT(W.M, i - 1, j - 1) + s; c[2] = AT(W.U, i - 1, j - 1) + s;
			st = first_max(&AT(W.M, i, j), c, 3);
			if (st >= 0) AT(W.pM, i, j) = (unsigned char)(st == 0 ? P_LOW : st == 1 ? P_MID : P_UPP);
			c[0] = AT(W.L, i - 1, j) + e; c[1] = AT(W.M, i - 1, j) + o;
			st = first_max(&AT(W.L, i, j), c, 2);
			if (st >= 0) AT(W.pL, i, j) = (unsigned char)(st == 0 ? P_LOW : P_MID);
			c[0] = AT(W.M, i, j - 1) + o; c[1] = AT(W.U, i, j - 1) + e;
			st = first_max(&AT(W.U, i, j), c, 2);
			if (st >= 0) AT(W.pU, i, j) = (unsigned char)(st == 0 ? P_MID : P_UPP);
		}
	/* final state :465-469 */
	c[0] = AT(W.L, l1, l2); c[1] = AT(W.M, l1, l2); c[2] = AT(W.U, l1, l2);
	st = first_max(score, c, 3);
	if (st < 0) { mats_free(&W); return -2; }
	*est = st == 0 ? ATO_ST_LOW : st == 1 ? ATO_ST_MID : ATO_ST_UPP;
	*ei = l1; *ej = l2;
	/* traceback :372-412 including the two padding loops :398-407 */
	rc = walk_affine(&W, l1, l2, st == 0 ? P_LOW : st == 1 ? P_MID : P_UPP, 1, b, &i, &j);
	if (!rc) {
		while (j > 0) { if (emit(b, ATO_OP_UPP)) { rc = -2; break; } --j; }
		while (!rc && i > 0) { if (emit(b, ATO_OP_LOW)) { rc = -2; break; } --i; }
	}
	mats_free(&W);
	return rc;
}

/* ---- local: align_local_affine alignment.h:805-847 ------------------------ */
static int do_local(const char *s1, int l1, const char *s2, int l2, const ato_scoring *sc,
                    double *score, int *ei, int *ej, int *est, opbuf *b)
{
	mats W; int i, j, st, rc, imax = -1, jmax = -1;
	double m = sc->m, u = sc->u, o = sc->o, e = sc->e, c[4], best = NEG;
	if (l1 < 1 || l2 < 1) return -2;        /* i_max/j_max uninitialised in the reference */
	if (mats_new(&W, l1, l2, 0)) { mats_free(&W); return -1; }
	/* no border init: calloc zeros stand for M, L and U */
	for (i = 1; i <= l1; ++i)
		for (j = 1; j <= l2; ++j) {
			double s = (s1[i - 1] == s2[j - 1]) ? m : u;
			c[0] = AT(W.L, i - 1, j - 1) + s; c[1] = AT(W.M, i - 1, j - 1) + s;
			c[2] = AT(W.U, i - 1, j - 1) + s; c[3] = 0.0;
			st = first_max(&AT(W.M, i, j), c, 4);
			AT(W.pM, i, j) = (unsigned char)(st == 0 ? P_LOW : st == 1 ? P_MID : st == 2 ? P_UPP : P_HOME);
			if (AT(W.M, i, j) > best) { best = AT(W.M, i, j); imax = i; jmax = j; }   /* :830-833 */
			c[0] = AT(W.L, i - 1, j) + e; c[1] = AT(W.M, i - 1, j) + o;
			st = first_max(&AT(W.L, i, j), c, 2);
			if (st >= 0) AT(W.pL, i, j) = (unsigned char)(st == 0 ? P_LOW : P_MID);
			c[0] = AT(W.M, i, j - 1) + o; c[1] = AT(W.U, i, j - 1) + e;
			st = first_max(&AT(W.U, i, j), c, 2);
			if (st >= 0) AT(W.pU, i, j) = (unsigned char)(st == 0 ? P_MID : P_UPP);
		}
	*score = best; *ei = imax; *ej = jmax; *est = ATO_ST_MID;
	rc = walk_affine(&W, imax, jmax, P_MID, 1, b, &i, &j);   /* :766-800 */
	mats_free(&W);
	return rc;
}

static int site_listed(int v, const int *a, int n)
{
	int k;
	for (k = 0; k < n; ++k) if (a[k] == v) return 1;
	return 0;
}

/* ---- fit (+jump): align_fit_affine_jump alignment.h:596-694 --------------- */
static int do_fit(const char *s1, int l1, const char *s2, int l2, const ato_scoring *sc,
                  double *score, int *ei, int *ej, int *est, opbuf *b)
{
	mats W; int i, j, st, rc, jmax = -1, state = 0, sj = sc->use_jump;
	double m = sc->m, u = sc->u, o = sc->o, e = sc->e, g = sc->j, c[4], best = NEG;
	if (l1 > l2) return -1;                  /* reference die()s :599 */
	if (l1 < 1) return -2;
	if (mats_new(&W, l1, l2, 1)) { mats_free(&W); return -1; }
	/* column 0 first, then row 0 overrides (0,0)  :612-624 */
	for (i = 0; i <= l1; ++i) { AT(W.M, i, 0) = NEG; AT(W.U, i, 0) = NEG; AT(W.L, i, 0) = NEG; AT(W.J, i, 0) = NEG; }
	for (j = 0; j <= l2; ++j) { AT(W.M, 0, j) = 0.0; AT(W.U, 0, j) = 0.0; AT(W.J, 0, j) = NEG; AT(W.L, 0, j) = NEG; }
	for (i = 1; i <= l1; ++i)
		for (j = 1; j <= l2; ++j) {
			double s = (s1[i - 1] == s2[j - 1]) ? m : u;
			c[0] = AT(W.L, i - 1, j - 1) + s; c[1] = AT(W.M, i - 1, j - 1) + s;
			c[2] = AT(W.U, i - 1, j - 1) + s; c[3] = AT(W.J, i - 1, j - 1) + s;
			st = first_max(&AT(W.M, i, j), c, sj ? 4 : 3);            /* :634-645 */
			if (st >= 0) AT(W.pM, i, j) = (unsigned char)(st == 0 ? P_LOW : st == 1 ? P_MID : st == 2 ? P_UPP : P_JUMP);
			c[0] = AT(W.L, i - 1, j) + e; c[1] = AT(W.M, i - 1, j) + o;
			st = first_max(&AT(W.L, i, j), c, 2);
			if (st >= 0) AT(W.pL, i, j) = (unsigned char)(st == 0 ? P_LOW : P_MID);
			c[0] = AT(W.M, i, j - 1) + o; c[1] = AT(W.U, i, j - 1) + e;
			st = first_max(&AT(W.U, i, j), c, 2);
			if (st >= 0) AT(W.pU, i, j) = (unsigned char)(st == 0 ? P_MID : P_UPP);
			if (sj) {
				/* :658-666.  isvalueinarray returns the enum `true`(0) when FOUND,
				 * and the caller tests C truthiness, so the M->J opening is
				 * allowed exactly when (j-1) is NOT a listed site. */
				if (!site_listed(j - 1, sc->sites, sc->nsites)) {
					c[0] = AT(W.M, i, j - 1) + g; c[1] = AT(W.J, i, j - 1);
					st = first_max(&AT(W.J, i, j), c, 2);
					if (st >= 0) AT(W.pJ, i, j) = (unsigned char)(st == 0 ? P_MID : P_JUMP);
				} else {
					c[0] = AT(W.J, i, j - 1);
					st = first_max(&AT(W.J, i, j), c, 1);
					if (st >= 0) AT(W.pJ, i, j) = P_JUMP;
				}
			}
		}
	/* end cell :673-690: M over j=0..l2-1 (strict <, first j), then L only if greater */
	for (j = 0; j < l2; ++j)
		if (best < AT(W.M, l1, j)) { best = AT(W.M, l1, j); jmax = j; state = P_MID; }
	for (j = 0; j < l2; ++j)
		if (best < AT(W.L, l1, j)) { best = AT(W.L, l1, j); jmax = j; state = P_LOW; }
	if (!state) { mats_free(&W); return -2; }
	*score = best; *ei = l1; *ej = jmax; *est = state == P_MID ? ATO_ST_MID : ATO_ST_LOW;
	rc = walk_affine(&W, l1, jmax, state, 0, b, &i, &j);       /* :558-592, while(i>0) */
	mats_free(&W);
	return rc;
}

/* ---- overlap: align_overlap alignment.h:926-964 --------------------------- */
static int do_overlap(const char *s1, int l1, const char *s2, int l2, const ato_scoring *sc,
                      double *score, int *ei, int *ej, int *est, opbuf *b)
{
	mats W; int i, j, st, jmax = -1, rc = 0;
	double m = sc->m, u = sc->u, o = sc->o, c[3], best = NEG;
	long guard;
	if (l2 < 1) return -2;                   /* j_max uninitialised */
	if (mats_new(&W, l1, l2, 0)) { mats_free(&W); return -1; }
	for (j = 0; j <= l2; ++j) AT(W.M, 0, j) = NEG;              /* :937 */
	for (i = 0; i <= l1; ++i) AT(W.M, i, 0) = 0.0;              /* :938 */
	for (i = 1; i <= l1; ++i)
		for (j = 1; j <= l2; ++j) {
			double s = (s1[i - 1] == s2[j - 1]) ? m : u;
			c[0] = AT(W.M, i, j - 1) + o; c[1] = AT(W.M, i - 1, j - 1) + s; c[2] = AT(W.M, i - 1, j) + o;
			st = first_max(&AT(W.M, i, j), c, 3);               /* linear gap; -e unused :944 */
			if (st >= 0) AT(W.pM, i, j) = (unsigned char)(st == 0 ? P_LEFT : st == 1 ? P_DIAG : P_RIGHT);
		}
	for (j = 0; j < l2; ++j)                                    /* :951-959 */
		if (best < AT(W.M, l1, j)) { best = AT(W.M, l1, j); jmax = j; }
	if (jmax < 0) { mats_free(&W); return -2; }
	*score = best; *ei = l1; *ej = jmax; *est = ATO_ST_MID;
	/* trace_back_overlap :896-922, while(j>0) */
	i = l1; j = jmax; guard = 4L * (l1 + l2) + 16;
	while (j > 0) {
		if (--guard < 0 || i < 0) { rc = -2; break; }
		switch (AT(W.pM, i, j)) {
		case P_LEFT:  if (emit(b, ATO_OP_UPP)) rc = -2; --j; break;
		case P_DIAG:  if (emit(b, ATO_OP_MID)) rc = -2; --i; --j; break;
		case P_RIGHT: if (emit(b, ATO_OP_LOW)) rc = -2; --i; break;
		default: rc = -2; break;
		}
		if (rc) break;
	}
	mats_free(&W);
	return rc;
}

/* ---- edit distance: edit_dist alignment.h:291-315 ------------------------- */
static int do_edit(const char *s1, int l1, const char *s2, int l2, const ato_scoring *sc, double *score)
{
	size_t cols = (size_t)l2 + 1;
	double *M = calloc((size_t)(l1 + 1) * cols, sizeof(double));
	double mism = sc->u;                      /* `-u` IS the (signed) mismatch cost; gap cost is 1 */
	int i, j;
	if (!M) return -1;
	for (i = 0; i <= l1; ++i) M[(size_t)i * cols] = i;
	for (j = 0; j <= l2; ++j) M[j] = j;
	for (i = 1; i <= l1; ++i)
		for (j = 1; j <= l2; ++j) {
			double a = M[(size_t)i * cols + j - 1] + 1;
			double d = M[(size_t)(i - 1) * cols + j - 1] + ((s1[i - 1] == s2[j - 1]) ? 0.0 : mism);
			double c = M[(size_t)(i - 1) * cols + j] + 1;
			double r = INFINITY;                 /* min3 :280-286 */
			if (a < r) r = a;
			if (d < r) r = d;
			if (c < r) r = c;
			M[(size_t)i * cols + j] = r;
		}
	*score = (double)(int)M[(size_t)l1 * cols + l2];
	free(M);
	return 0;
}

int ato_align(int mode, const char *s1, int l1, const char *s2, int l2,
              const ato_scoring *sc, double *score,
              char *r1, char *r2, int cap, int *rlen,
              int *end_i, int *end_j, int *start_state,
              unsigned char *ops, int *nops)
{
	opbuf b; int rc, ei = 0, ej = 0, est = 0;
	unsigned char *own = NULL;
	if (!s1 || !s2 || !sc || !score || l1 < 0 || l2 < 0) return -1;
	if (mode != ATO_EDIT && (!r1 || !r2 || cap < l1 + l2 + 1)) return -1;
	b.n = 0; b.cap = l1 + l2 + 1;
	b.ops = ops ? ops : (own = malloc((size_t)b.cap));
	if (!b.ops) return -1;
	switch (mode) {
	case ATO_GLOBAL:  rc = do_global(s1, l1, s2, l2, sc, score, &ei, &ej, &est, &b); break;
	case ATO_LOCAL:   rc = do_local(s1, l1, s2, l2, sc, score, &ei, &ej, &est, &b); break;
	case ATO_FIT:     rc = do_fit(s1, l1, s2, l2, sc, score, &ei, &ej, &est, &b); break;
	case ATO_OVERLAP: rc = do_overlap(s1, l1, s2, l2, sc, score, &ei, &ej, &est, &b); break;
	case ATO_EDIT:    rc = do_edit(s1, l1, s2, l2, sc, score); break;
	default: rc = -1;
	}
	if (!rc && mode != ATO_EDIT) {
		render(b.ops, b.n, s1, s2, ei, ej, r1, r2);
		if (rlen) *rlen = b.n;
	} else if (rlen) *rlen = 0;
	if (end_i) *end_i = ei;
	if (end_j) *end_j = ej;
	if (start_state) *start_state = est;
	if (nops) *nops = rc ? 0 : b.n;
	free(own);
	return rc;
}

double ato_time_batch(int mode, int n, const char *blob, int l1, int l2,
                      const ato_scoring *sc, double *checksum)
{
	struct timespec t0, t1;
	int cap = l1 + l2 + 1, rl = 0, k;
	char *r1 = malloc((size_t)cap), *r2 = malloc((size_t)cap);
	double acc = 0, s = 0;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (k = 0; k < n; ++k) {
		const char *p = blob + (size_t)k * (size_t)(l1 + l2);
		ato_align(mode, p, l1, p + l1, l2, sc, &s, r1, r2, cap, &rl, NULL, NULL, NULL, NULL, NULL);
		acc += s + rl;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(r1); free(r2);
	*checksum = acc;
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
