/*
 * cli.c -- `alignTools <command> [options] <target.fa>`: the reference's command line
 * (src/main.c:16-57 and the five main_* drivers of src/alignment.h) in front of the
 * MI355X shim.  Option strings, defaults, usage texts (typos included), stdout/stderr
 * bytes and return codes follow the reference; the align_*() calls land on the GPU.
 *
 * Extension (absent from the reference, does not change the five commands):
 *   alignTools batch <command> [options] <pairs.fa>
 *     records (2k, 2k+1) of the file form pair k; all pairs go to the GPU in one batch.
 */
#define _POSIX_C_SOURCE 200809L
#include "at_host.h"
#include <dirent.h>
#include <math.h>
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#define PACKAGE_VERSION "0.7.23-r15"

static int usage(void)
{
	fprintf(stderr, "\n");
	fprintf(stderr, "Program: alignTools (pairwise DNA sequence alignment)\n");
	fprintf(stderr, "Version: %s\n", PACKAGE_VERSION);
	fprintf(stderr, "Contact: Rongxin Fang <r3fang@ucsd.edu>\n\n");
	fprintf(stderr, "Usage:   alignTools <command> [options]\n\n");
	fprintf(stderr, "Command: global     global (needle) alignment allows affine gap\n");
	fprintf(stderr, "         local      smith-waterman with affine gap\n");
	fprintf(stderr, "         fit        fit alingment allows affine gap plus jump state\n");
	fprintf(stderr, "         overlap    overlap alignment\n");
	fprintf(stderr, "         edit       edit distance\n");
	fprintf(stderr, "\n");
	return 1;
}

enum { C_GLOBAL, C_LOCAL, C_FIT, C_OVERLAP, C_EDIT };
static const char *cmd_name[] = {"global", "local", "fit", "overlap", "edit"};

/* getopt loops of alignment.h:323-331, 481-489, 703-713, 856-864, 971-979 */
static int parse_opts(int cmd, int argc, char *argv[], opt_t *opt)
{
	int c;
	const char *spec = cmd == C_EDIT ? "m:u:o:e" : "m:u:o:e:j:s";
	while ((c = getopt(argc, argv, spec)) >= 0) {
		switch (c) {
		case 'm': opt->m = atoi(optarg); break;
		case 'u': opt->u = atoi(optarg); break;
		case 'o': opt->o = atoi(optarg); break;
		case 'e':
			if (!optarg) return 1;   /* edit declares "-e" without an argument and then calls atoi(NULL): reject instead of crashing */
			opt->e = atoi(optarg); break;
		case 'j': if (cmd != C_FIT) return 1; opt->j = atoi(optarg); break;
		case 's': if (cmd != C_FIT) return 1; opt->s = AT_TRUE; break;
		default: return 1;
		}
	}
	return 0;
}

static void cmd_usage(int cmd, const opt_t *opt)
{
	fprintf(stderr, "\n");
	fprintf(stderr, "Usage:   alignTools %s [options] <target.fa>\n\n", cmd_name[cmd]);
	if (cmd == C_EDIT) {
		fprintf(stderr, "Options: -u INT   mismatch penalty [%d]\n", opt->u);
		fprintf(stderr, "         -o INT   gap penalty [%d]\n", opt->o);
	} else {
		fprintf(stderr, "Options: -m INT   score for a match [%d]\n", opt->m);
		fprintf(stderr, "         -u INT   mismatch penalty [%d]\n", opt->u);
		fprintf(stderr, "         -o INT   gap open penalty [%d]\n", opt->o);
		fprintf(stderr, "         -e INT   gap extension penalty [%d]\n", opt->e);
		if (cmd == C_FIT) {
			fprintf(stderr, "         -j INT   jump penality [%d]\n", opt->j);
			fprintf(stderr, "         -s       weather jump state include\n");
		}
	}
	fprintf(stderr, "\n");
}

static int main_single(int cmd, int argc, char *argv[])
{
	opt_t *opt = init_opt();
	kstring_t *ks1, *ks2, *r1, *r2;
	if (parse_opts(cmd, argc, argv, opt)) { free(opt); return 1; }
	if (optind + 1 > argc) { cmd_usage(cmd, opt); free(opt); return 1; }
	ks1 = (kstring_t *)at_xmalloc(sizeof(kstring_t)); memset(ks1, 0, sizeof *ks1);
	ks2 = (kstring_t *)at_xmalloc(sizeof(kstring_t)); memset(ks2, 0, sizeof *ks2);
	/* overlap reads argv[1], not argv[argc-1] (alignment.h:994): after getopt's permutation any
	 * option makes that an option string -> "Can't open -m" */
	kstring_read(cmd == C_OVERLAP ? argv[1] : argv[argc - 1], ks1, ks2, opt);
	if (ks1->s == NULL || ks2->s == NULL) die("fail to read sequence\n");
	if (cmd == C_EDIT) {
		printf("edit_distance=%d\n", edit_dist(ks1, ks2, opt));
		kstring_destory(ks1); kstring_destory(ks2);
		free(opt->sites.pos); free(opt);
		return 0;
	}
	if (cmd == C_FIT && ks1->l > ks2->l) die("first sequence must be shorter than the second\n");   /* :731 */
	r1 = (kstring_t *)at_xmalloc(sizeof(kstring_t)); memset(r1, 0, sizeof *r1);
	r2 = (kstring_t *)at_xmalloc(sizeof(kstring_t)); memset(r2, 0, sizeof *r2);
	r1->s = (char *)at_xmalloc(ks1->l + ks2->l + 1); memset(r1->s, 0, ks1->l + ks2->l + 1);
	r2->s = (char *)at_xmalloc(ks1->l + ks2->l + 1); memset(r2->s, 0, ks1->l + ks2->l + 1);
	switch (cmd) {
	case C_GLOBAL: printf("score=%f\n", align_gla(ks1, ks2, r1, r2, opt)); break;
	case C_LOCAL: printf("score=%f\n", align_local_affine(ks1, ks2, r1, r2, opt)); break;
	case C_FIT: printf("score=%f\n", align_fit_affine_jump(ks1, ks2, r1, r2, opt)); break;
	default: printf("%f\n", align_overlap(ks1, ks2, r1, r2, opt)); break;   /* no "score=" prefix, :1000 */
	}
	printf("%s\n%s\n", r1->s, r2->s);
	kstring_destory(ks1); kstring_destory(ks2); kstring_destory(r1); kstring_destory(r2);
	free(opt->sites.pos); free(opt);
	return 0;
}

/* ---- batch extension: N pairs per file ----
 *   alignTools batch <command> [options] [--score-only] [--all-vs-all] [--gpus N] <pairs.fa>
 *     default        records (2k, 2k+1) form pair k
 *     --all-vs-all   the records are reads; every ordered pair a < b is aligned as s1 = read a, s2 = read b (not for fit)
 *     --score-only   no tracebacks: one line per pair (name1, name2, score)
 *     --min-score T  (overlap --all-vs-all --score-only) only pairs that score at least T are printed; pairs the bit-parallel
 *                    bound proves below T are not swept at all (at_set_min_score)
 *     --gpus N       one process per GPU: this process starts N workers (itself, with AT_RANK / AT_WORLD / AT_DEVICE /
 *                    AT_COMM_DIR in their environment), rank 0's options are broadcast over RCCL, every rank aligns a
 *                    contiguous share of the pairs on its own GPU, results are gathered over RCCL and rank 0 prints them
 *                    in pair order.  The output is that of --gpus 1.
 * One process streams: the file is parsed block-wise in chunks of pairs (fasta.c) on the main thread while a second
 * thread -- which also pays the HIP start-up, in the shadow of the first chunks' parsing -- sends each chunk to the GPU
 * and writes its results with one fwrite.  Memory is bounded by the chunks in flight, whatever the size of the file;
 * --all-vs-all keeps the read set and streams slices of the triangle instead (at_align_allpairs_stream). */
typedef struct { int score_only, all_vs_all, gpus, min_on, min_score; } batch_flags;

/* linear index p of the strict upper triangle of n x n (row-major) -> (a, b), a < b: closed form + integer correction */
static void tri_seek(int64_t p, int64_t n, int64_t *a, int64_t *b)
{
	const double d = (double)(2 * n - 1);
	int64_t r = (int64_t)((d - sqrt(d * d - 8.0 * (double)p)) * 0.5);
	if (r < 0) r = 0;
	if (r > n - 2) r = n - 2;
	while (r > 0 && r * (2 * n - r - 1) / 2 > p) --r;
	while ((r + 1) * (2 * n - r - 2) / 2 <= p) ++r;
	*a = r; *b = p - r * (2 * n - r - 1) / 2 + r + 1;
}
#define TRI_NEXT(a, b, n) do { if (++(b) >= (n)) { ++(a); (b) = (a) + 1; } } while (0)

/* ---- output text of a chunk, written with one fwrite ---- */
typedef struct { char *s; size_t l, m; } tbuf;
static void tb_need(tbuf *t, size_t extra)
{
	if (t->l + extra <= t->m) return;
	while (t->m < t->l + extra) t->m = t->m ? t->m * 2 : (size_t)1 << 20;
	t->s = (char *)at_xrealloc(t->s, t->m);
}
static void tb_put(tbuf *t, const char *s, size_t n) { tb_need(t, n); memcpy(t->s + t->l, s, n); t->l += n; }
/* "score=%f" of an integer-valued double / "edit_distance=%d", without printf */
static void tb_score(tbuf *t, int32_t v, int is_edit)
{
	char d[16];
	int n = 0;
	uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
	tb_need(t, 48);
	if (is_edit) { memcpy(t->s + t->l, "edit_distance=", 14); t->l += 14; }
	else { memcpy(t->s + t->l, "score=", 6); t->l += 6; }
	if (v < 0) t->s[t->l++] = '-';
	do { d[n++] = (char)('0' + u % 10); u /= 10; } while (u);
	while (n) t->s[t->l++] = d[--n];
	if (!is_edit) { memcpy(t->s + t->l, ".000000", 7); t->l += 7; }
}
/* "<name1>\t<name2>\t<score>\n[<r1>\n<r2>\n]" */
static void tb_pair(tbuf *t, const char *na, const char *nb, int32_t score, int is_edit, const char *r1, const char *r2, size_t rl)
{
	const size_t la = strlen(na), lb = strlen(nb);
	tb_need(t, la + lb + 2 * rl + 64);
	memcpy(t->s + t->l, na, la); t->l += la; t->s[t->l++] = '\t';
	memcpy(t->s + t->l, nb, lb); t->l += lb; t->s[t->l++] = '\t';
	tb_score(t, score, is_edit);
	t->s[t->l++] = '\n';
	if (r1) {
		memcpy(t->s + t->l, r1, rl); t->l += rl; t->s[t->l++] = '\n';
		memcpy(t->s + t->l, r2, rl); t->l += rl; t->s[t->l++] = '\n';
	}
}
static void tb_flush(tbuf *t)
{
	if (t->l && fwrite(t->s, 1, t->l, stdout) != t->l) die("write error on stdout");
	t->l = 0;
}

/* ---- a slice of pairs: descriptors in, results out; buffers are reused from slice to slice ---- */
typedef struct {
	int64_t cap;
	int64_t *off1, *off2, *slot;
	int32_t *l1, *l2, *score, *ei, *ej, *st, *nops;
	char *r1, *r2;
	size_t rcap;
} slice_t;

static void slice_reserve(slice_t *w, int64_t n)
{
	if (n <= w->cap) return;
	w->cap = n + n / 4 + 16;
	w->off1 = (int64_t *)at_xrealloc(w->off1, (size_t)w->cap * 8); w->off2 = (int64_t *)at_xrealloc(w->off2, (size_t)w->cap * 8);
	w->slot = (int64_t *)at_xrealloc(w->slot, (size_t)w->cap * 8);
	w->l1 = (int32_t *)at_xrealloc(w->l1, (size_t)w->cap * 4); w->l2 = (int32_t *)at_xrealloc(w->l2, (size_t)w->cap * 4);
	w->score = (int32_t *)at_xrealloc(w->score, (size_t)w->cap * 4); w->ei = (int32_t *)at_xrealloc(w->ei, (size_t)w->cap * 4);
	w->ej = (int32_t *)at_xrealloc(w->ej, (size_t)w->cap * 4); w->st = (int32_t *)at_xrealloc(w->st, (size_t)w->cap * 4);
	w->nops = (int32_t *)at_xrealloc(w->nops, (size_t)w->cap * 4);
}
static void slice_free(slice_t *w)
{
	free(w->off1); free(w->off2); free(w->slot); free(w->l1); free(w->l2); free(w->score); free(w->ei); free(w->ej); free(w->st);
	free(w->nops); free(w->r1); free(w->r2);
	memset(w, 0, sizeof *w);
}
/* off1 / l1 / off2 / l2 of pairs 0 .. n-1 are filled in: align them (strings rendered on the GPU when tb) */
static void slice_run(slice_t *w, at_handle *h, int cmd, int tb, const uint8_t *blob, int64_t n)
{
	const int mode = cmd == C_GLOBAL ? AT_MODE_GLOBAL : cmd == C_LOCAL ? AT_MODE_LOCAL : cmd == C_FIT ? AT_MODE_FIT
	               : cmd == C_OVERLAP ? AT_MODE_OVERLAP : AT_MODE_EDIT;
	int64_t k, sl = 0;
	int rc;
	for (k = 0; k < n; ++k) {
		if (cmd == C_FIT && w->l1[k] > w->l2[k]) die("first sequence must be shorter than the second\n");
		w->slot[k] = sl; sl += (int64_t)w->l1[k] + w->l2[k] + 1;
	}
	if (n <= 0) return;
	if (tb) {
		if ((size_t)sl + 64 > w->rcap) {
			w->rcap = (size_t)sl + (size_t)sl / 4 + 64;
			free(w->r1); free(w->r2);
			w->r1 = (char *)at_xmalloc(w->rcap); w->r2 = (char *)at_xmalloc(w->rcap);
		}
		rc = at_align_batch_strings(h, mode, n, blob, w->off1, w->l1, w->off2, w->l2, w->score, w->ei, w->ej, w->st, w->r1, w->r2, w->slot, w->nops);
	} else
		rc = at_align_batch(h, mode, n, blob, w->off1, w->l1, w->off2, w->l2, 0, w->score, w->ei, w->ej, w->st, NULL, NULL, NULL);
	if (rc != AT_OK) die("%s", at_last_error(h));
}

/* AT_CLI_TRACE=1: milliseconds since the first call, on stderr, at the stages of a batch run (where does the wall clock go?) */
static void trace(const char *what, long long n)
{
	static int on = -1;
	static struct timespec t0;
	struct timespec t;
	if (on < 0) { on = getenv("AT_CLI_TRACE") != NULL; clock_gettime(CLOCK_MONOTONIC, &t0); }
	if (!on) return;
	clock_gettime(CLOCK_MONOTONIC, &t);
	fprintf(stderr, "[trace] %8.2f ms  %s %lld\n", (double)(t.tv_sec - t0.tv_sec) * 1e3 + (double)(t.tv_nsec - t0.tv_nsec) * 1e-6, what, n);
}

static int64_t env_i64(const char *name, int64_t dflt)
{
	const char *v = getenv(name);
	return v && *v ? (int64_t)atoll(v) : dflt;
}
/* a knob that is a step of a loop or a read size: zero or negative would loop forever or read as end of file */
static int64_t env_pos(const char *name, int64_t dflt)
{
	const int64_t v = env_i64(name, dflt);
	return v >= 1 ? v : dflt;
}

static void set_sites(opt_t *opt, const at_chunk *c)
{
	if (opt->s != AT_TRUE) return;
	if (c->n < 2 || c->comment_off[1] == (size_t)-1) die("fail to read junction sites");
	opt->sites.size = (size_t)at_parse_sites(c->comments + c->comment_off[1], &opt->sites.pos);
}

static void *gpu_warmup(void *unused) { (void)unused; (void)at_host_handle(); return NULL; }

/* ---- pair lists in one process: reader (main thread) -> ring of chunks -> GPU + output (second thread) ---- */
#define RING 4
typedef struct {
	pthread_mutex_t mu;
	pthread_cond_t cv;
	at_chunk chunk[RING];
	int filled[RING];            /* 1: parsed, waiting for the GPU thread */
	int head, tail, closed;      /* the reader fills `tail`, the GPU thread takes `head` */
	int cmd, tb;
	opt_t *opt;
} pipe_t;

static void *pipe_consumer(void *arg)
{
	pipe_t *pp = (pipe_t *)arg;
	at_handle *h = (trace("gpu thread starts", 0), at_host_handle());                /* HIP start-up: in the shadow of the reader */
	slice_t w;
	tbuf out = {NULL, 0, 0};
	const opt_t *opt = pp->opt;
	int rc = at_set_scoring(h, opt->m, opt->u, opt->o, opt->e, opt->j, opt->s == AT_TRUE, opt->sites.pos, (int)opt->sites.size);
	if (rc != AT_OK) die("%s", at_last_error(h));
	memset(&w, 0, sizeof w);
	trace("gpu handle ready", 0);
	for (;;) {
		at_chunk *c;
		int64_t n, k;
		pthread_mutex_lock(&pp->mu);
		while (!pp->filled[pp->head] && !pp->closed) pthread_cond_wait(&pp->cv, &pp->mu);
		if (!pp->filled[pp->head]) { pthread_mutex_unlock(&pp->mu); break; }
		c = &pp->chunk[pp->head];
		pthread_mutex_unlock(&pp->mu);
		n = (int64_t)(c->n / 2);
		slice_reserve(&w, n);
		for (k = 0; k < n; ++k) {
			w.off1[k] = (int64_t)c->off[2 * k]; w.l1[k] = (int32_t)c->len[2 * k];
			w.off2[k] = (int64_t)c->off[2 * k + 1]; w.l2[k] = (int32_t)c->len[2 * k + 1];
		}
		slice_run(&w, h, pp->cmd, pp->tb, c->blob, n);
		trace("chunk aligned, pairs", n);
		for (k = 0; k < n; ++k)
			tb_pair(&out, c->names + c->name_off[2 * k], c->names + c->name_off[2 * k + 1], w.score[k], pp->cmd == C_EDIT,
			        pp->tb ? w.r1 + w.slot[k] : NULL, pp->tb ? w.r2 + w.slot[k] : NULL, pp->tb ? (size_t)w.nops[k] : 0);
		tb_flush(&out);
		trace("chunk written", n);
		pthread_mutex_lock(&pp->mu);
		pp->filled[pp->head] = 0;
		pp->head = (pp->head + 1) % RING;
		pthread_cond_broadcast(&pp->cv);
		pthread_mutex_unlock(&pp->mu);
	}
	slice_free(&w);
	free(out.s);
	return NULL;
}

static int batch_stream_pairs(int cmd, opt_t *opt, int tb, at_reader *rd)
{
	pipe_t *pp = (pipe_t *)at_xmalloc(sizeof *pp);
	pthread_t th;
	int started = 0, q;
	size_t total = 0;
	const size_t first_pairs = (size_t)env_pos("AT_CLI_FIRST_CHUNK", 8192), chunk_pairs = (size_t)env_pos("AT_CLI_CHUNK", 32768);
	const size_t max_bases = (size_t)env_pos("AT_CLI_CHUNK_BASES", (int64_t)256 << 20);
	memset(pp, 0, sizeof *pp);
	pthread_mutex_init(&pp->mu, NULL);
	pthread_cond_init(&pp->cv, NULL);
	pp->cmd = cmd; pp->tb = tb; pp->opt = opt;
	for (;;) {
		at_chunk *c;
		size_t got;
		pthread_mutex_lock(&pp->mu);
		while (pp->filled[pp->tail]) pthread_cond_wait(&pp->cv, &pp->mu);   /* every slot is waiting for the GPU: the reader pauses */
		c = &pp->chunk[pp->tail];
		pthread_mutex_unlock(&pp->mu);
		at_chunk_reset(c);
		got = at_reader_read(rd, 2 * (started ? chunk_pairs : first_pairs), max_bases, c);
		if (got & 1) got += at_reader_read(rd, 1, (size_t)-1, c);          /* (the bases limit fell between the two records of a pair) */
		total += got;
		trace("chunk parsed, records", (long long)got);
		if (!started) {
			/* the first chunk decides what a small file's errors are, before any thread or GPU exists */
			if (total < 2 || (got & 1)) die("batch input needs an even number of records (got %d)", (int)total);
			set_sites(opt, c);
		}
		if (got == 0) break;
		if (got & 1) {          /* an odd record at the very end of a file of several chunks */
			pthread_mutex_lock(&pp->mu); pp->closed = 1; pthread_cond_broadcast(&pp->cv); pthread_mutex_unlock(&pp->mu);
			pthread_join(th, NULL);
			die("batch input needs an even number of records (got %d)", (int)total);
		}
		pthread_mutex_lock(&pp->mu);
		pp->filled[pp->tail] = 1;
		pp->tail = (pp->tail + 1) % RING;
		pthread_cond_broadcast(&pp->cv);
		pthread_mutex_unlock(&pp->mu);
		if (!started) {
			if (pthread_create(&th, NULL, pipe_consumer, pp) != 0) die("cannot start the GPU thread");
			started = 1;
		}
	}
	pthread_mutex_lock(&pp->mu); pp->closed = 1; pthread_cond_broadcast(&pp->cv); pthread_mutex_unlock(&pp->mu);
	if (started) pthread_join(th, NULL);
	for (q = 0; q < RING; ++q) at_chunk_free(&pp->chunk[q]);
	pthread_mutex_destroy(&pp->mu); pthread_cond_destroy(&pp->cv);
	free(pp);
	return 0;
}

/* ---- everything else keeps the whole record set: --all-vs-all, and the ranks of --gpus N ---- */
typedef struct { const at_chunk *c; int64_t a, b, nrec; int is_edit; tbuf out; int min_on, min_score; int32_t *keep; int64_t keep_first; } ava_print;

static int print_slice(void *user, int64_t first, int64_t n, const int32_t *score, const int32_t *ei, const int32_t *ej, const int32_t *st)
{
	ava_print *ap = (ava_print *)user;
	int64_t k;
	(void)first; (void)ei; (void)ej; (void)st;
	if (ap->keep) {                     /* a rank of many: its share of the scores, for the gather */
		memcpy(ap->keep + (first - ap->keep_first), score, (size_t)n * 4);
		return 0;
	}
	for (k = 0; k < n; ++k) {
		if (!ap->min_on || score[k] >= ap->min_score)   /* (a pair the filter stopped carries an upper bound below the threshold) */
			tb_pair(&ap->out, ap->c->names + ap->c->name_off[ap->a], ap->c->names + ap->c->name_off[ap->b], score[k], ap->is_edit, NULL, NULL, 0);
		TRI_NEXT(ap->a, ap->b, ap->nrec);
		if (ap->out.l > ((size_t)8 << 20)) tb_flush(&ap->out);
	}
	tb_flush(&ap->out);
	return 0;
}

static int batch_worker(int cmd, opt_t *opt, const batch_flags *bf, const char *fname, int rank, int world, const char *comm_dir)
{
	at_reader *rd = at_reader_open(fname);
	at_chunk c;
	at_handle *h;
	pthread_t warm;
	slice_t w;
	tbuf out = {NULL, 0, 0};
	int rc;
	int64_t nrec, npairs, lo, hi, a = 0, b = 0;
	int64_t *roff = NULL;
	int32_t *rlen = NULL;
	const int tb = !bf->score_only && cmd != C_EDIT;
	const int comm = world > 1 || getenv("AT_COMM_FORCE_RCCL") != NULL;   /* (the latter: the RCCL calls at world size 1, for tests) */
	const int mode = cmd == C_GLOBAL ? AT_MODE_GLOBAL : cmd == C_LOCAL ? AT_MODE_LOCAL : cmd == C_FIT ? AT_MODE_FIT
	               : cmd == C_OVERLAP ? AT_MODE_OVERLAP : AT_MODE_EDIT;
	const int64_t chunk_pairs = env_pos("AT_CLI_CHUNK", 32768) * 8;
	if (!rd) die("Can't open %s\n", fname);
	if (bf->all_vs_all && cmd == C_FIT) die("--all-vs-all: fit needs ordered pairs (first sequence shorter than the second)");
	if (!bf->all_vs_all && !comm) {
		rc = batch_stream_pairs(cmd, opt, tb, rd);
		at_reader_close(rd);
		return rc;
	}
	memset(&c, 0, sizeof c);
	memset(&w, 0, sizeof w);
	/* the first records decide whether there is anything to do; then HIP starts up beside the rest of the parsing */
	(void)at_reader_read(rd, 4, (size_t)-1, &c);
	if (bf->all_vs_all && c.n < 2) die("--all-vs-all needs at least two records (got %d)", (int)c.n);
	if (!bf->all_vs_all && c.n < 2) die("batch input needs an even number of records (got %d)", (int)c.n);
	set_sites(opt, &c);
	if (pthread_create(&warm, NULL, gpu_warmup, NULL) != 0) die("cannot start the GPU thread");
	while (at_reader_read(rd, (size_t)-1, (size_t)-1, &c) > 0) {}
	at_reader_close(rd);
	pthread_join(warm, NULL);
	nrec = (int64_t)c.n;
	if (!bf->all_vs_all && (nrec & 1)) die("batch input needs an even number of records (got %d)", (int)nrec);
	npairs = bf->all_vs_all ? nrec * (nrec - 1) / 2 : nrec / 2;
	/* this rank's contiguous share of the pairs: rank r owns [ceil(n r / N), ceil(n (r + 1) / N)) */
	lo = (npairs * rank + world - 1) / world;
	hi = (npairs * (rank + 1) + world - 1) / world;
	if (hi > npairs) hi = npairs;
	h = at_host_handle();
	rc = at_set_scoring(h, opt->m, opt->u, opt->o, opt->e, opt->j, opt->s == AT_TRUE, opt->sites.pos, (int)opt->sites.size);
	if (rc == AT_OK && comm) {
		trace("comm: init, rank", rank);
		rc = at_comm_init(h, rank, world, comm_dir);
		trace("comm: communicator up, world", world);
		if (rc == AT_OK) rc = at_comm_broadcast_scoring(h);      /* rank 0's options are everybody's */
		trace("comm: scoring broadcast done", rc);
	}
 	if (rc == AT_OK && bf->min_on) rc = at_set_min_score(h, 1, bf->min_score);
	if (rc != AT_OK) die("%s", at_last_error(h));
	if (bf->all_vs_all) {
		int64_t k;
		roff = (int64_t *)at_xmalloc((size_t)nrec * 8); rlen = (int32_t *)at_xmalloc((size_t)nrec * 4);
		for (k = 0; k < nrec; ++k) { roff[k] = (int64_t)c.off[k]; rlen[k] = (int32_t)c.len[k]; }
	}
	if (!comm && !tb) {
		/* scores of the whole triangle: the reads go up once, the pairs are enumerated on the GPU, slices are printed as they arrive */
		ava_print ap;
		memset(&ap, 0, sizeof ap);
		ap.c = &c; ap.nrec = nrec; ap.is_edit = cmd == C_EDIT; ap.a = 0; ap.b = 1;
		ap.min_on = bf->min_on; ap.min_score = bf->min_score;
		rc = at_align_allpairs_stream(h, mode, nrec, c.blob, roff, rlen, 0, npairs, env_i64("AT_CLI_CHUNK", 0), print_slice, &ap);
		if (rc != AT_OK) die("%s", at_last_error(h));
		free(ap.out.s);
	} else {
		/* slices of this rank's share; one process prints every slice, a rank of many collects its share for the gather */
		int32_t *myscore = comm ? (int32_t *)at_xmalloc((size_t)(hi - lo + 1) * 4) : NULL;
		tbuf pay = {NULL, 0, 0};
		int64_t s0, k;
		if (bf->all_vs_all && hi > lo) tri_seek(lo, nrec, &a, &b);
		const int streamed = bf->all_vs_all && !tb && comm;
		if (streamed && hi > lo) {
			/* this rank's share of the triangle's scores: the reads go up once, the slices land in myscore (one
			 * at_align_allpairs call per slice uploaded and packed the whole read set again every time) */
			ava_print ap;
			memset(&ap, 0, sizeof ap);
			ap.keep = myscore; ap.keep_first = lo;
			rc = at_align_allpairs_stream(h, mode, nrec, c.blob, roff, rlen, lo, hi - lo, env_i64("AT_CLI_CHUNK", 0), print_slice, &ap);
			if (rc != AT_OK) die("%s", at_last_error(h));
		}
		for (s0 = lo; s0 < hi && !streamed; s0 += chunk_pairs) {
			const int64_t n = hi - s0 < chunk_pairs ? hi - s0 : chunk_pairs;
			int64_t a0 = a, b0 = b;
			slice_reserve(&w, n);
			if (bf->all_vs_all && !tb) {
				rc = at_align_allpairs(h, mode, nrec, c.blob, roff, rlen, s0, n, 0, w.score, w.ei, w.ej, w.st, NULL, NULL, NULL);
				if (rc != AT_OK) die("%s", at_last_error(h));
				for (k = 0; k < n; ++k) TRI_NEXT(a, b, nrec);
			} else {
				for (k = 0; k < n; ++k) {
					const int64_t ra = bf->all_vs_all ? a : 2 * (s0 + k), rb = bf->all_vs_all ? b : 2 * (s0 + k) + 1;
					w.off1[k] = (int64_t)c.off[ra]; w.l1[k] = (int32_t)c.len[ra]; w.off2[k] = (int64_t)c.off[rb]; w.l2[k] = (int32_t)c.len[rb];
					if (bf->all_vs_all) TRI_NEXT(a, b, nrec);
				}
				slice_run(&w, h, cmd, tb, c.blob, n);
			}
			if (!comm) {
				for (k = 0; k < n; ++k) {
					const int64_t ra = bf->all_vs_all ? a0 : 2 * (s0 + k), rb = bf->all_vs_all ? b0 : 2 * (s0 + k) + 1;
					if (!bf->min_on || tb || w.score[k] >= bf->min_score)
					tb_pair(&out, c.names + c.name_off[ra], c.names + c.name_off[rb], w.score[k], cmd == C_EDIT,
					        tb ? w.r1 + w.slot[k] : NULL, tb ? w.r2 + w.slot[k] : NULL, tb ? (size_t)w.nops[k] : 0);
					if (bf->all_vs_all) TRI_NEXT(a0, b0, nrec);
				}
				tb_flush(&out);
			} else {
				memcpy(myscore + (s0 - lo), w.score, (size_t)n * 4);
				if (tb)
					for (k = 0; k < n; ++k) {   /* every pair's two strings back to back with their terminators */
						tb_put(&pay, w.r1 + w.slot[k], (size_t)w.nops[k] + 1);
						tb_put(&pay, w.r2 + w.slot[k], (size_t)w.nops[k] + 1);
					}
			}
		}
		if (comm) {
			/* gather: the scores (4 bytes per pair) and, with tracebacks, the strings -- at_comm_allgather sends the sizes
			 * first, then one padded payload; rank 0 prints everything in pair order */
			int64_t *bytes = (int64_t *)at_xmalloc((size_t)world * 8), o = 0, q;
			int32_t *allscore = (int32_t *)at_xmalloc((size_t)(npairs + 1) * 4);
			char *allpay = NULL;
			trace("comm: aligned my share, pairs", hi - lo);
			rc = at_comm_allgather(h, myscore, (hi - lo) * 4, allscore, npairs * 4, bytes);
			trace("comm: scores gathered, bytes", npairs * 4);
			if (rc == AT_OK && tb) {
				int64_t mine = (int64_t)pay.l, total = 0, *sizes = (int64_t *)at_xmalloc((size_t)world * 8);
				rc = at_comm_allgather(h, &mine, 8, sizes, (int64_t)world * 8, bytes);          /* how much to expect in all */
				for (q = 0; rc == AT_OK && q < world; ++q) total += sizes[q];
				allpay = (char *)at_xmalloc((size_t)total + 1);
				if (rc == AT_OK) rc = at_comm_allgather(h, pay.s, mine, allpay, total, bytes);
				trace("comm: strings gathered, bytes", total);
				free(sizes);
			}
			if (rc != AT_OK) die("%s", at_last_error(h));
			if (rank == 0) {
				a = 0; b = 1;
				for (q = 0; q < npairs; ++q) {
					const int64_t ra = bf->all_vs_all ? a : 2 * q, rb = bf->all_vs_all ? b : 2 * q + 1;
					const char *x = tb ? allpay + o : NULL;
					const size_t xl = tb ? strlen(x) : 0;
					if (!bf->min_on || tb || allscore[q] >= bf->min_score)
						tb_pair(&out, c.names + c.name_off[ra], c.names + c.name_off[rb], allscore[q], cmd == C_EDIT, x, tb ? x + xl + 1 : NULL, xl);
					if (tb) o += 2 * ((int64_t)xl + 1);
					if (bf->all_vs_all) TRI_NEXT(a, b, nrec);
					if (out.l > ((size_t)8 << 20)) tb_flush(&out);
				}
				tb_flush(&out);
			}
			at_comm_destroy(h);
			free(bytes); free(allscore); free(allpay);
		}
		free(myscore); free(pay.s);
	}
	slice_free(&w);
	free(out.s); free(roff); free(rlen);
	at_chunk_free(&c);
	return 0;
}

/* start one worker per GPU (this binary again, told its rank through the environment) and wait for them */
static int batch_launch(int world, char *argv0, int argc, char *argv[])
{
	char dir[] = "/tmp/alignTools.XXXXXX", buf[32];
	pid_t *pid = (pid_t *)at_xmalloc((size_t)world * sizeof(pid_t));
	int r, status, worst = 0, live = 0;
	char **av = (char **)at_xmalloc((size_t)(argc + 2) * sizeof(char *));
	if (!mkdtemp(dir)) die("cannot create a rendezvous directory under /tmp");
	av[0] = argv0;
	for (r = 0; r < argc; ++r) av[r + 1] = argv[r];
	av[argc + 1] = NULL;
	fflush(stdout); fflush(stderr);
	for (r = 0; r < world; ++r) {
		pid[r] = fork();
		if (pid[r] < 0) die("fork failed");
		if (pid[r] == 0) {
			snprintf(buf, sizeof buf, "%d", r); setenv("AT_RANK", buf, 1);
			if (!getenv("AT_ONE_DEVICE")) setenv("AT_DEVICE", buf, 1);      /* rank r on GPU r (AT_ONE_DEVICE: rehearsals on one card) */
			snprintf(buf, sizeof buf, "%d", world); setenv("AT_WORLD", buf, 1);
			setenv("AT_COMM_DIR", dir, 1);
			execv("/proc/self/exe", av);
			_exit(127);
		}
		++live;
	}
	/* a rank that fails leaves the others waiting in a collective: the first failure ends the ranks that are still running
	 * (pids of ranks already reaped are forgotten, so that a recycled pid is never signalled) */
	while (live > 0) {
		const pid_t done = waitpid(-1, &status, 0);
		int q, code = done < 0 || !WIFEXITED(status) ? 255 : WEXITSTATUS(status);
		if (done < 0) break;
		for (q = 0; q < world; ++q) if (pid[q] == done) { pid[q] = 0; --live; }
		if (code > worst) worst = code;
		if (code != 0)
			for (q = 0; q < world; ++q) if (pid[q] > 0) kill(pid[q], SIGTERM);
	}
	/* the rendezvous directory held a few small files */
	{
		DIR *d = opendir(dir);
		if (d) {
			struct dirent *e;
			char path[512];
			while ((e = readdir(d)) != NULL) {
				if (strcmp(e->d_name, ".") == 0 || strcmp(e->d_name, "..") == 0) continue;
				snprintf(path, sizeof path, "%s/%s", dir, e->d_name);
				(void)unlink(path);
			}
			closedir(d);
		}
		(void)rmdir(dir);
	}
	free(pid); free(av);
	return worst;
}

static int main_batch(int argc, char *argv[], char *argv0)
{
	int cmd = -1, k, n = 0;
	opt_t *opt = init_opt();
	batch_flags bf = {0, 0, 1, 0, 0};
	char **av = (char **)at_xmalloc((size_t)(argc + 1) * sizeof(char *));
	const char *usage_line = "Usage:   alignTools batch <global|local|fit|overlap|edit> [options] [--score-only] [--all-vs-all] [--min-score T] [--gpus N] <pairs.fa>\n";
	/* the long flags of the extension are taken out before getopt sees the reference's short options */
	for (k = 0; k < argc; ++k) {
		if (strcmp(argv[k], "--score-only") == 0) bf.score_only = 1;
		else if (strcmp(argv[k], "--all-vs-all") == 0) bf.all_vs_all = 1;
		else if (strcmp(argv[k], "--gpus") == 0 && k + 1 < argc) bf.gpus = atoi(argv[++k]);
		else if (strcmp(argv[k], "--min-score") == 0 && k + 1 < argc) { bf.min_on = 1; bf.min_score = atoi(argv[++k]); }
		else av[n++] = argv[k];
	}
	av[n] = NULL;
	if (n < 2 || bf.gpus < 1 || bf.gpus > 64) { fprintf(stderr, "%s", usage_line); free(opt); free(av); return 1; }
	for (k = 0; k < 5; ++k) if (strcmp(av[1], cmd_name[k]) == 0) cmd = k;
	if (cmd < 0) { fprintf(stderr, "[main] unrecognized command '%s'\n", av[1]); free(opt); free(av); return 1; }
	if (bf.min_on && !(cmd == C_OVERLAP && bf.all_vs_all && bf.score_only)) {
		fprintf(stderr, "--min-score goes with `batch overlap --all-vs-all --score-only`\n%s", usage_line); free(opt); free(av); return 1;
	}
	if (parse_opts(cmd, n - 1, av + 1, opt)) { free(opt); free(av); return 1; }
	if (optind + 1 > n - 1) { cmd_usage(cmd, opt); free(opt); free(av); return 1; }
	if (bf.gpus > 1 && !getenv("AT_RANK")) {        /* the launcher: never touches a GPU itself */
		k = batch_launch(bf.gpus, argv0, argc, argv);
		free(opt); free(av);
		return k ? k : -1;                          /* -1: the workers printed everything, the [main] trailer too */
	}
	{
		const int rank = getenv("AT_RANK") ? atoi(getenv("AT_RANK")) : 0, world = getenv("AT_WORLD") ? atoi(getenv("AT_WORLD")) : 1;
		k = batch_worker(cmd, opt, &bf, av[n - 1], rank, world, getenv("AT_COMM_DIR") ? getenv("AT_COMM_DIR") : "/tmp");
		free(opt->sites.pos); free(opt); free(av);
		return k == 0 && rank != 0 ? -1 : k;        /* only rank 0 prints the [main] trailer */
	}
}

/* A process that has used the GPU ends without the HIP runtime's tear-down (a tenth of a second on the MI355X box -- a third of
 * a 100 000-pair batch run): everything is flushed, the kernel driver reclaims the rest.  AT_FAST_EXIT=0 returns normally. */
static int leave(int ret)
{
	const char *v = getenv("AT_FAST_EXIT");
	if (!at_host_handle_exists() || (v && *v == '0')) return ret;
	if ((fflush(stdout) != 0 || ferror(stdout)) && ret == 0) ret = 1;   /* (a final write that failed must not leave with 0) */
	fflush(stderr);
	_exit(ret);
}

int main(int argc, char *argv[])
{
	int i, ret, cmd = -1, k;
	trace("main", 0);
	if (argc < 2) return usage();
	for (k = 0; k < 5; ++k) if (strcmp(argv[1], cmd_name[k]) == 0) cmd = k;
	if (cmd >= 0) ret = main_single(cmd, argc - 1, argv + 1);
	else if (strcmp(argv[1], "batch") == 0) {
		ret = main_batch(argc - 1, argv + 1, argv[0]);
		if (ret == -1) return leave(0);   /* a worker other than rank 0, or the launcher: nothing more to say */
	}
	else {
		fprintf(stderr, "[main] unrecognized command '%s'\n", argv[1]);
		return 1;
	}
	if (ret == 0) {
		fflush(stdout);
		fprintf(stderr, "[%s] Version: %s\n", __func__, PACKAGE_VERSION);
		fprintf(stderr, "[%s] CMD:", __func__);
		for (i = 0; i < argc; ++i) fprintf(stderr, " %s", argv[i]);
		fprintf(stderr, "\n");
	}
	trace("leaving main", ret);
	return leave(ret);
}
