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
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
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

/* ---- batch extension: N pairs per file, one GPU batch per process ----
 *   alignTools batch <command> [options] [--score-only] [--all-vs-all] [--gpus N] <pairs.fa>
 *     default        records (2k, 2k+1) form pair k
 *     --all-vs-all   the records are reads; every ordered pair a < b is aligned as s1 = read a, s2 = read b (not for fit)
 *     --score-only   no tracebacks: one line per pair (name1, name2, score)
 *     --gpus N       one process per GPU: this process starts N workers (itself, with AT_RANK / AT_WORLD / AT_DEVICE /
 *                    AT_COMM_DIR in their environment), rank 0's options are broadcast over RCCL, every rank aligns a
 *                    contiguous share of the pairs on its own GPU, results are gathered over RCCL and rank 0 prints them
 *                    in pair order.  The output is that of --gpus 1. */
typedef struct { int score_only, all_vs_all, gpus; } batch_flags;

/* linear index p of the strict upper triangle of n x n (row-major) -> (a, b), a < b */
static void tri_pair(int64_t p, int64_t n, int64_t *a, int64_t *b)
{
	int64_t r = 0, before = 0;
	while (before + (n - 1 - r) <= p) { before += n - 1 - r; ++r; }
	*a = r; *b = r + 1 + (p - before);
}

static int batch_worker(int cmd, opt_t *opt, const batch_flags *bf, const char *fname, int rank, int world, const char *comm_dir)
{
	at_records rec;
	at_handle *h;
	int mode, rc;
	size_t nrec, p, tot = 0;
	int64_t npairs, lo, hi, n, k;
	uint8_t *blob;
	int64_t *roff, *off1, *off2, *slot;
	int32_t *rlen, *l1, *l2, *score, *ei, *ej, *st, *nops;
	char *r1 = NULL, *r2 = NULL;
	const int tb = !bf->score_only && cmd != C_EDIT;
	if (at_read_records(fname, &rec) != 0) die("Can't open %s\n", fname);
	nrec = rec.n;
	if (bf->all_vs_all) {
		if (cmd == C_FIT) die("--all-vs-all: fit needs ordered pairs (first sequence shorter than the second)");
		if (nrec < 2) die("--all-vs-all needs at least two records (got %d)", (int)nrec);
		npairs = (int64_t)nrec * ((int64_t)nrec - 1) / 2;
	} else {
		if (nrec < 2 || (nrec & 1)) die("batch input needs an even number of records (got %d)", (int)nrec);
		npairs = (int64_t)nrec / 2;
	}
	if (opt->s == AT_TRUE) {
		if (rec.comment[1] == NULL) die("fail to read junction sites");
		opt->sites.size = (size_t)at_parse_sites(rec.comment[1], &opt->sites.pos);
	}
	for (p = 0; p < nrec; ++p) tot += rec.len[p];
	blob = (uint8_t *)at_xmalloc(tot + 1);
	roff = (int64_t *)at_xmalloc(nrec * 8); rlen = (int32_t *)at_xmalloc(nrec * 4);
	tot = 0;
	for (p = 0; p < nrec; ++p) {
		roff[p] = (int64_t)tot; rlen[p] = (int32_t)rec.len[p];
		memcpy(blob + tot, rec.seq[p], rec.len[p]); tot += rec.len[p];
	}
	/* this rank's contiguous share of the pairs: rank r owns [ceil(n r / N), ceil(n (r + 1) / N)) */
	lo = (npairs * rank + world - 1) / world;
	hi = (npairs * (rank + 1) + world - 1) / world;
	if (hi > npairs) hi = npairs;
	n = hi - lo;
	off1 = (int64_t *)at_xmalloc((size_t)(n + 1) * 8); off2 = (int64_t *)at_xmalloc((size_t)(n + 1) * 8); slot = (int64_t *)at_xmalloc((size_t)(n + 1) * 8);
	l1 = (int32_t *)at_xmalloc((size_t)(n + 1) * 4); l2 = (int32_t *)at_xmalloc((size_t)(n + 1) * 4);
	score = (int32_t *)at_xmalloc((size_t)(n + 1) * 4); ei = (int32_t *)at_xmalloc((size_t)(n + 1) * 4); ej = (int32_t *)at_xmalloc((size_t)(n + 1) * 4);
	st = (int32_t *)at_xmalloc((size_t)(n + 1) * 4); nops = (int32_t *)at_xmalloc((size_t)(n + 1) * 4);
	{
		int64_t sl = 0, a = 0, b = 0;
		for (k = 0; k < n; ++k) {
			if (bf->all_vs_all) tri_pair(lo + k, (int64_t)nrec, &a, &b); else { a = 2 * (lo + k); b = a + 1; }
			off1[k] = roff[a]; l1[k] = rlen[a]; off2[k] = roff[b]; l2[k] = rlen[b];
			slot[k] = sl; sl += (int64_t)l1[k] + l2[k] + 1;
			if (cmd == C_FIT && l1[k] > l2[k]) die("first sequence must be shorter than the second\n");
		}
		if (tb) { r1 = (char *)at_xmalloc((size_t)sl + 64); r2 = (char *)at_xmalloc((size_t)sl + 64); }
	}
	mode = cmd == C_GLOBAL ? AT_MODE_GLOBAL : cmd == C_LOCAL ? AT_MODE_LOCAL : cmd == C_FIT ? AT_MODE_FIT
	     : cmd == C_OVERLAP ? AT_MODE_OVERLAP : AT_MODE_EDIT;
	h = at_host_handle();
	rc = at_set_scoring(h, opt->m, opt->u, opt->o, opt->e, opt->j, opt->s == AT_TRUE, opt->sites.pos, (int)opt->sites.size);
	const int comm = world > 1 || getenv("AT_COMM_FORCE_RCCL") != NULL;   /* (the latter: the RCCL calls at world size 1, for tests) */
	if (rc == AT_OK && comm) {
		rc = at_comm_init(h, rank, world, comm_dir);
		if (rc == AT_OK) rc = at_comm_broadcast_scoring(h);      /* rank 0's options are everybody's */
	}
	if (rc == AT_OK && n > 0) {
		if (bf->all_vs_all && !tb)   /* scores of a slice of the triangle: the reads go up once, the pairs are enumerated on the GPU */
			rc = at_align_allpairs(h, mode, (int64_t)nrec, blob, roff, rlen, lo, n, 0, score, ei, ej, st, NULL, NULL, NULL);
		else if (!tb)
			rc = at_align_batch(h, mode, n, blob, off1, l1, off2, l2, 0, score, ei, ej, st, NULL, NULL, NULL);
		else        /* strings are rendered on the GPU (at_render.hip.h) */
			rc = at_align_batch_strings(h, mode, n, blob, off1, l1, off2, l2, score, ei, ej, st, r1, r2, slot, nops);
	}
	if (rc != AT_OK) die("%s", at_last_error(h));
	if (!comm) {
		int64_t a = 0, b = 0;
		for (k = 0; k < n; ++k) {
			if (bf->all_vs_all) tri_pair(lo + k, (int64_t)nrec, &a, &b); else { a = 2 * (lo + k); b = a + 1; }
			if (cmd == C_EDIT) printf("%s\t%s\tedit_distance=%d\n", rec.name[a], rec.name[b], score[k]);
			else if (!tb) printf("%s\t%s\tscore=%f\n", rec.name[a], rec.name[b], (double)score[k]);
			else printf("%s\t%s\tscore=%f\n%s\n%s\n", rec.name[a], rec.name[b], (double)score[k], r1 + slot[k], r2 + slot[k]);
		}
	} else {
		/* gather: the scores (4 bytes per pair) and, with tracebacks, every pair's two strings back to back with their
		 * terminators -- at_comm_allgather sends the sizes first, then one padded payload; rank 0 prints */
		int64_t *bytes = (int64_t *)at_xmalloc((size_t)world * 8), paylen = 0, o, q;
		int32_t *allscore = (int32_t *)at_xmalloc((size_t)(npairs + 1) * 4);
		char *pay = NULL, *allpay = NULL;
		rc = at_comm_allgather(h, score, n * 4, allscore, npairs * 4, bytes);
		if (rc == AT_OK && tb) {
			int64_t cap = 0;
			for (p = 0; p < nrec; ++p) cap += (int64_t)rec.len[p];
			for (k = 0; k < n; ++k) paylen += 2 * ((int64_t)nops[k] + 1);
			pay = (char *)at_xmalloc((size_t)paylen + 1);
			for (k = 0, o = 0; k < n; ++k) {
				memcpy(pay + o, r1 + slot[k], (size_t)nops[k] + 1); o += nops[k] + 1;
				memcpy(pay + o, r2 + slot[k], (size_t)nops[k] + 1); o += nops[k] + 1;
			}
			/* an upper bound of everything: every pair's strings are at most l1 + l2 + 1 long, twice */
			cap = bf->all_vs_all ? 2 * ((int64_t)(nrec - 1) * cap + npairs) : 2 * (cap + npairs);
			allpay = (char *)at_xmalloc((size_t)cap + 1);
			rc = at_comm_allgather(h, pay, paylen, allpay, cap, bytes);
		}
		if (rc != AT_OK) die("%s", at_last_error(h));
		if (rank == 0) {
			int64_t a = 0, b = 0;
			for (q = 0, o = 0; q < npairs; ++q) {
				if (bf->all_vs_all) tri_pair(q, (int64_t)nrec, &a, &b); else { a = 2 * q; b = a + 1; }
				if (cmd == C_EDIT) printf("%s\t%s\tedit_distance=%d\n", rec.name[a], rec.name[b], allscore[q]);
				else if (!tb) printf("%s\t%s\tscore=%f\n", rec.name[a], rec.name[b], (double)allscore[q]);
				else {
					const char *x = allpay + o, *y = x + strlen(x) + 1;
					printf("%s\t%s\tscore=%f\n%s\n%s\n", rec.name[a], rec.name[b], (double)allscore[q], x, y);
					o += 2 * ((int64_t)strlen(x) + 1);
				}
			}
		}
		at_comm_destroy(h);
		free(bytes); free(allscore); free(pay); free(allpay);
	}
	free(blob); free(roff); free(rlen); free(off1); free(off2); free(slot);
	free(l1); free(l2); free(score); free(ei); free(ej); free(st); free(nops); free(r1); free(r2);
	at_free_records(&rec);
	return 0;
}

/* start one worker per GPU (this binary again, told its rank through the environment) and wait for them */
static int batch_launch(int world, char *argv0, int argc, char *argv[])
{
	char dir[] = "/tmp/alignTools.XXXXXX", buf[32];
	pid_t *pid = (pid_t *)at_xmalloc((size_t)world * sizeof(pid_t));
	int r, status, worst = 0;
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
	}
	/* a rank that fails leaves the others waiting in a collective: the first failure ends them all */
	for (r = 0; r < world; ++r) {
		const pid_t done = waitpid(-1, &status, 0);
		int q, code = done < 0 || !WIFEXITED(status) ? 255 : WEXITSTATUS(status);
		if (code > worst) worst = code;
		if (code != 0) {
			for (q = 0; q < world; ++q) if (pid[q] != done) kill(pid[q], SIGTERM);
		}
	}
	/* the rendezvous directory held a few small files */
	{
		char cmdline[96];
		snprintf(cmdline, sizeof cmdline, "rm -rf %s", dir);
		if (system(cmdline) != 0) worst = worst ? worst : 0;
	}
	free(pid); free(av);
	return worst;
}

static int main_batch(int argc, char *argv[], char *argv0)
{
	int cmd = -1, k, n = 0;
	opt_t *opt = init_opt();
	batch_flags bf = {0, 0, 1};
	char **av = (char **)at_xmalloc((size_t)(argc + 1) * sizeof(char *));
	const char *usage_line = "Usage:   alignTools batch <global|local|fit|overlap|edit> [options] [--score-only] [--all-vs-all] [--gpus N] <pairs.fa>\n";
	/* the long flags of the extension are taken out before getopt sees the reference's short options */
	for (k = 0; k < argc; ++k) {
		if (strcmp(argv[k], "--score-only") == 0) bf.score_only = 1;
		else if (strcmp(argv[k], "--all-vs-all") == 0) bf.all_vs_all = 1;
		else if (strcmp(argv[k], "--gpus") == 0 && k + 1 < argc) bf.gpus = atoi(argv[++k]);
		else av[n++] = argv[k];
	}
	av[n] = NULL;
	if (n < 2 || bf.gpus < 1 || bf.gpus > 64) { fprintf(stderr, "%s", usage_line); free(opt); free(av); return 1; }
	for (k = 0; k < 5; ++k) if (strcmp(av[1], cmd_name[k]) == 0) cmd = k;
	if (cmd < 0) { fprintf(stderr, "[main] unrecognized command '%s'\n", av[1]); free(opt); free(av); return 1; }
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

int main(int argc, char *argv[])
{
	int i, ret, cmd = -1, k;
	if (argc < 2) return usage();
	for (k = 0; k < 5; ++k) if (strcmp(argv[1], cmd_name[k]) == 0) cmd = k;
	if (cmd >= 0) ret = main_single(cmd, argc - 1, argv + 1);
	else if (strcmp(argv[1], "batch") == 0) {
		ret = main_batch(argc - 1, argv + 1, argv[0]);
		if (ret == -1) return 0;        /* a worker other than rank 0, or the launcher: nothing more to say */
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
	return ret;
}
