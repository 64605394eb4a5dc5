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

at_handle *at_host_handle(void)
{
	static at_handle *h = NULL;
	if (!h) {
		const char *dev = getenv("AT_DEVICE");
		int id = dev ? atoi(dev) : 0;
		int rc = at_init(dev ? &id : NULL, dev ? 1 : 0, &h);
		if (rc != AT_OK) die("%s", at_last_error(NULL));
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

static double run_pair(int mode, kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, opt_t *opt)
{
	at_handle *h = at_host_handle();
	const int64_t off1 = 0, off2 = (int64_t)s1->l, opsoff = 0;
	const int32_t l1 = (int32_t)s1->l, l2 = (int32_t)s2->l;
	int32_t score = 0, ei = 0, ej = 0, st = 0, nops = 0;
	size_t tot = s1->l + s2->l;
	uint8_t *blob = (uint8_t *)malloc(tot + 1), *ops = (uint8_t *)malloc(tot + 64);
	int rc;
	if (!blob || !ops) die("mycalloc failure requesting %d of size %d bytes", (int)tot, 1);
	memcpy(blob, s1->s, s1->l);
	memcpy(blob + s1->l, s2->s, s2->l);
	rc = at_set_scoring(h, opt->m, opt->u, opt->o, opt->e, opt->j, opt->s == AT_TRUE, opt->sites.pos, (int)opt->sites.size);
	if (rc == AT_OK)
		rc = at_align_batch(h, mode, 1, blob, &off1, &l1, &off2, &l2, mode != AT_MODE_EDIT, &score, &ei, &ej, &st,
		                    ops, &opsoff, &nops);
	if (rc == AT_ERR_FIT_ORDER) die("first sequence must be shorter than the second to do fitting alignment");   /* :599 */
	if (rc != AT_OK) die("%s", at_last_error(h));
	if (mode != AT_MODE_EDIT) {
		char *a = (char *)malloc((size_t)nops + 1), *b = (char *)malloc((size_t)nops + 1);
		if (!a || !b) die("mycalloc failure requesting %d of size %d bytes", nops + 1, 1);
		if (at_render(ops, nops, (const uint8_t *)s1->s, ei, (const uint8_t *)s2->s, ej, a, b) != AT_OK)
			die("internal error: traceback inconsistent with the sequences");
		replace(r1, a, (size_t)nops);
		replace(r2, b, (size_t)nops);
		free(a); free(b);
	}
	free(blob); free(ops);
	return (double)score;
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
