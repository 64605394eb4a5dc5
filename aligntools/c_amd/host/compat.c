/*
 * compat.c -- the reference's function surface (include/aligntools.h) on top of
 * the gfx950 shim.  C host code only marshals: one pair in, at_align_batch on
 * the GPU, ops rendered into the two gapped strings.  No DP cell is computed
 * on the CPU.
 */
#define _POSIX_C_SOURCE 200809L
#include "at_host.h"
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int g_fit_debug = -1;

void at_set_fit_debug_line(int on) { g_fit_debug = on ? 1 : 0; }

/* alignment.h:69-79 */
void die(const char *format, ...)
{
	va_list args;
	va_start(args, format);
	fprintf(stderr, "FATAL ERROR: ");
	vfprintf(stderr, format, args);
	fprintf(stderr, "\n");
	va_end(args);
	exit(-1);
}

/* alignment.h:102-114 */
opt_t *init_opt(void)
{
	opt_t *opt = (opt_t *)calloc(1, sizeof(opt_t));
	if (!opt) die("mycalloc failure requesting %d of size %d bytes", 1, (int)sizeof(opt_t));
	opt->o = -5; opt->e = -1; opt->m = 1; opt->u = -2; opt->j = -10;
	opt->s = AT_FALSE;
	opt->sites.size = 0;
	opt->sites.pos = NULL;
	return opt;
}

void kstring_destory(kstring_t *ks)
{
	free(ks->s);
	free(ks);
}

static at_handle *g_handle = NULL;
int at_host_handle_exists(void) { return g_handle != NULL; }

at_handle *at_host_handle(void)
{
	at_handle *h = g_handle;
	if (!h) {
		const char *dev = getenv("AT_DEVICE");
		int id = dev ? atoi(dev) : 0;
		int rc = at_init(dev ? &id : NULL, dev ? 1 : 0, &h);
		if (rc != AT_OK) die("%s", at_last_error(NULL));
		g_handle = h;
	}
	return h;
}

static void replace(kstring_t *r, const char *s, size_t n)
{
	free(r->s);
	r->s = (char *)malloc(n + 1);
	if (!r->s) die("mycalloc failure requesting %d of size %d bytes", (int)n + 1, 1);
	memcpy(r->s, s, n);
	r->s[n] = 0;
	r->l = n;
	r->m = n + 1;
}

/* what a fill leaves behind (include/aligntools.h: matrix_t): the pointer matrix itself stays on the GPU */
struct at_matrix {
	int mode;                 /* AT_MODE_* */
	int32_t score, ei, ej, st, nops;
	uint8_t *ops;             /* END -> START */
	size_t l1, l2;
};

static matrix_t *fill_pair(int mode, kstring_t *s1, kstring_t *s2, opt_t *opt)
{
	at_handle *h = at_host_handle();
	const int64_t off1 = 0, off2 = (int64_t)s1->l, opsoff = 0;
	const int32_t l1 = (int32_t)s1->l, l2 = (int32_t)s2->l;
	size_t tot = s1->l + s2->l;
	matrix_t *S = (matrix_t *)at_xmalloc(sizeof *S);
	uint8_t *blob = (uint8_t *)at_xmalloc(tot + 1);
	int rc;
	memset(S, 0, sizeof *S);
	S->mode = mode; S->l1 = s1->l; S->l2 = s2->l;
	S->ops = (uint8_t *)at_xmalloc(tot + 64);
	memcpy(blob, s1->s, s1->l);
	memcpy(blob + s1->l, s2->s, s2->l);
	rc = at_set_scoring(h, opt->m, opt->u, opt->o, opt->e, opt->j, opt->s == AT_TRUE, opt->sites.pos, (int)opt->sites.size);
	if (rc == AT_OK)
		rc = at_align_batch(h, mode, 1, blob, &off1, &l1, &off2, &l2, mode != AT_MODE_EDIT, &S->score, &S->ei, &S->ej, &S->st,
		                    S->ops, &opsoff, &S->nops);
	if (rc == AT_ERR_FIT_ORDER) die("first sequence must be shorter than the second to do fitting alignment");   /* :599 */
	if (rc != AT_OK) die("%s", at_last_error(h));
	free(blob);
	return S;
}

static void walk_into(const matrix_t *S, kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2)
{
	char *a = (char *)at_xmalloc((size_t)S->nops + 1), *b = (char *)at_xmalloc((size_t)S->nops + 1);
	if (s1->l != S->l1 || s2->l != S->l2 ||
	    at_render(S->ops, S->nops, (const uint8_t *)s1->s, S->ei, (const uint8_t *)s2->s, S->ej, a, b) != AT_OK)
		die("internal error: traceback inconsistent with the sequences");
	replace(r1, a, (size_t)S->nops);
	replace(r2, b, (size_t)S->nops);
	free(a); free(b);
}

void destory_matrix(matrix_t *S)
{
	if (!S) return;
	free(S->ops);
	free(S);
}

static double run_pair(int mode, kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, opt_t *opt)
{
	matrix_t *S = fill_pair(mode, s1, s2, opt);
	const double score = (double)S->score;
	if (mode != AT_MODE_EDIT) walk_into(S, s1, s2, r1, r2);
	destory_matrix(S);
	return score;
}

/* ---- the fill and the four trace_back_*() as calls of their own (include/aligntools.h) ---- */
static int ref_state(int st) { return st == AT_ST_LOW ? AT_LOW : st == AT_ST_UPP ? AT_UPP : AT_MID; }

matrix_t *at_fill_matrix(int fill_mode, kstring_t *s1, kstring_t *s2, opt_t *opt, double *score, int *state, int *i, int *j)
{
	matrix_t *S;
	if (s1 == NULL || s2 == NULL || opt == NULL) die("align: parameter error\n");
	if (fill_mode < AT_FILL_GLOBAL || fill_mode > AT_FILL_OVERLAP) die("at_fill_matrix: unknown mode %d", fill_mode);
	if (fill_mode == AT_FILL_FIT && s1->l > s2->l) die("first sequence must be shorter than the second to do fitting alignment");   /* :599 */
	S = fill_pair(fill_mode, s1, s2, opt);          /* (AT_FILL_* = AT_MODE_* for the four modes with a traceback) */
	if (score) *score = (double)S->score;
	if (state) *state = ref_state(S->st);
	if (i) *i = S->ei;
	if (j) *j = S->ej;
	return S;
}

static void walk_checked(const char *who, int mode, matrix_t *S, kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2,
                         int check_state, int state, int check_cell, int i, int j)
{
	if (S == NULL || s1 == NULL || s2 == NULL || r1 == NULL || r2 == NULL) die("%s: parameter error\n", who);
	if (S->mode != mode) die("%s: this matrix was filled by another alignment mode", who);
	if ((check_state && state != ref_state(S->st)) || (check_cell && (i != S->ei || j != S->ej)))
		die("%s: only the traceback from the fill's own end cell exists (state %d, cell %d,%d): the pointer matrix stays on the GPU",
		    who, ref_state(S->st), S->ei, S->ej);
	walk_into(S, s1, s2, r1, r2);
}

void trace_back_gla(matrix_t *S, kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, int state)                       /* :372 */
{
	walk_checked("trace_back_gla", AT_MODE_GLOBAL, S, s1, s2, r1, r2, 1, state, 0, 0, 0);
}
void trace_back_fit_affine_jump(matrix_t *S, kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, int state, int i, int j)   /* :558 */
{
	walk_checked("trace_back_fit_affine_jump", AT_MODE_FIT, S, s1, s2, r1, r2, 1, state, 1, i, j);
}
void trace_back_local_affine(matrix_t *S, kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, int i, int j)             /* :766 */
{
	walk_checked("trace_back_local_affine", AT_MODE_LOCAL, S, s1, s2, r1, r2, 0, 0, 1, i, j);
}
void trace_back_overlap(matrix_t *S, kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, int i, int j)                  /* :896 */
{
	walk_checked("trace_back_overlap", AT_MODE_OVERLAP, S, s1, s2, r1, r2, 0, 0, 1, i, j);
}

double align_gla(kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, opt_t *opt)
{
	if (s1 == NULL || s2 == NULL || r1 == NULL || r2 == NULL) die("align: parameter error\n");   /* :419 */
	return run_pair(AT_MODE_GLOBAL, s1, s2, r1, r2, opt);
}

double align_local_affine(kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, opt_t *opt)
{
	if (s1 == NULL || s2 == NULL || r1 == NULL || r2 == NULL) die("align: parameter error\n");   /* :807 */
	return run_pair(AT_MODE_LOCAL, s1, s2, r1, r2, opt);
}

double align_fit_affine_jump(kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, opt_t *opt)
{
	if (s1 == NULL || s2 == NULL || r1 == NULL || r2 == NULL || opt == NULL) die("align: parameter error\n");   /* :598 */
	if (s1->l > s2->l) die("first sequence must be shorter than the second to do fitting alignment");          /* :599 */
	if (g_fit_debug < 0) g_fit_debug = getenv("AT_QUIET_FIT") && atoi(getenv("AT_QUIET_FIT")) ? 0 : 1;
	if (g_fit_debug) printf("asDAsdaSDAsdasDAsdaSD\n");   /* the reference's stray debug line, :602 */
	return run_pair(AT_MODE_FIT, s1, s2, r1, r2, opt);
}

double align_overlap(kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, opt_t *opt)
{
	if (s1 == NULL || s2 == NULL || r1 == NULL || r2 == NULL) die("align_overlap: parameter error\n");   /* :927 */
	return run_pair(AT_MODE_OVERLAP, s1, s2, r1, r2, opt);
}

int edit_dist(kstring_t *s1, kstring_t *s2, opt_t *opt)
{
	if (s1 == NULL || s2 == NULL || opt == NULL) die("edit_dist: parameter error\n");   /* :293 */
	return (int)run_pair(AT_MODE_EDIT, s1, s2, NULL, NULL, opt);
}

/* alignment.h:217-262 */
void kstring_read(char *fname, kstring_t *str1, kstring_t *str2, opt_t *opt)
{
	at_records rec;
	if (fname == NULL || str1 == NULL || str2 == NULL || opt == NULL) die("kstring_read: input error");
	if (at_read_records(fname, &rec) != 0) die("Can't open %s\n", fname);
	if (rec.n > 2) die("input fasta file has more than 2 sequences");
	if (rec.n < 2) die("read_kstring: fail to read sequence");
	str1->s = at_xstrdup(rec.seq[0]); str1->l = strlen(str1->s);
	str2->s = at_xstrdup(rec.seq[1]); str2->l = strlen(str2->s);
	if (opt->s == AT_TRUE) {
		if (rec.comment[1] == NULL) die("fail to read junction sites");
		printf("%s\n", rec.comment[1]);                                  /* :249 */
		opt->sites.size = (size_t)at_parse_sites(rec.comment[1], &opt->sites.pos);
	}
	at_free_records(&rec);
}
