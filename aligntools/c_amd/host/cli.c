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
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

/* ---- batch extension: N pairs per file, one GPU batch ---- */
static int main_batch(int argc, char *argv[])
{
	int cmd = -1, k, mode, rc;
	opt_t *opt = init_opt();
	at_records rec;
	at_handle *h;
	size_t n, p, tot = 0;
	uint8_t *blob;
	char *r1, *r2;
	int64_t *off1, *off2, *stroff;
	int32_t *l1, *l2, *score, *ei, *ej, *st, *nops;
	if (argc < 2) { fprintf(stderr, "Usage:   alignTools batch <global|local|fit|overlap|edit> [options] <pairs.fa>\n"); free(opt); return 1; }
	for (k = 0; k < 5; ++k) if (strcmp(argv[1], cmd_name[k]) == 0) cmd = k;
	if (cmd < 0) { fprintf(stderr, "[main] unrecognized command '%s'\n", argv[1]); free(opt); return 1; }
	if (parse_opts(cmd, argc - 1, argv + 1, opt)) { free(opt); return 1; }
	if (optind + 1 > argc - 1) { cmd_usage(cmd, opt); free(opt); return 1; }
	if (at_read_records(argv[argc - 1], &rec) != 0) die("Can't open %s\n", argv[argc - 1]);
	if (rec.n < 2 || (rec.n & 1)) die("batch input needs an even number of records (got %d)", (int)rec.n);
	n = rec.n / 2;
	if (opt->s == AT_TRUE) {
		if (rec.comment[1] == NULL) die("fail to read junction sites");
		opt->sites.size = (size_t)at_parse_sites(rec.comment[1], &opt->sites.pos);
	}
	for (p = 0; p < rec.n; ++p) tot += rec.len[p];
	blob = (uint8_t *)at_xmalloc(tot + 1);
	r1 = (char *)at_xmalloc(tot + n + 64); r2 = (char *)at_xmalloc(tot + n + 64);
	off1 = (int64_t *)at_xmalloc(n * 8); off2 = (int64_t *)at_xmalloc(n * 8); stroff = (int64_t *)at_xmalloc(n * 8);
	l1 = (int32_t *)at_xmalloc(n * 4); l2 = (int32_t *)at_xmalloc(n * 4); score = (int32_t *)at_xmalloc(n * 4);
	ei = (int32_t *)at_xmalloc(n * 4); ej = (int32_t *)at_xmalloc(n * 4); st = (int32_t *)at_xmalloc(n * 4); nops = (int32_t *)at_xmalloc(n * 4);
	tot = 0;
	for (p = 0; p < n; ++p) {
		off1[p] = (int64_t)tot; l1[p] = (int32_t)rec.len[2 * p];
		memcpy(blob + tot, rec.seq[2 * p], rec.len[2 * p]); tot += rec.len[2 * p];
		off2[p] = (int64_t)tot; l2[p] = (int32_t)rec.len[2 * p + 1];
		memcpy(blob + tot, rec.seq[2 * p + 1], rec.len[2 * p + 1]); tot += rec.len[2 * p + 1];
		stroff[p] = off1[p] + (int64_t)p;          /* slots of l1+l2+1 bytes */
		if (cmd == C_FIT && l1[p] > l2[p]) die("first sequence must be shorter than the second\n");
	}
	mode = cmd == C_GLOBAL ? AT_MODE_GLOBAL : cmd == C_LOCAL ? AT_MODE_LOCAL : cmd == C_FIT ? AT_MODE_FIT
	     : cmd == C_OVERLAP ? AT_MODE_OVERLAP : AT_MODE_EDIT;
	h = at_host_handle();
	rc = at_set_scoring(h, opt->m, opt->u, opt->o, opt->e, opt->j, opt->s == AT_TRUE, opt->sites.pos, (int)opt->sites.size);
	if (rc == AT_OK)   /* strings are rendered on the GPU (at_render.hip.h) */
		rc = cmd == C_EDIT ? at_align_batch(h, mode, (int64_t)n, blob, off1, l1, off2, l2, 0, score, ei, ej, st, NULL, NULL, NULL)
		                   : at_align_batch_strings(h, mode, (int64_t)n, blob, off1, l1, off2, l2, score, ei, ej, st, r1, r2, stroff, nops);
	if (rc != AT_OK) die("%s", at_last_error(h));
	for (p = 0; p < n; ++p) {
		if (cmd == C_EDIT) { printf("%s\t%s\tedit_distance=%d\n", rec.name[2 * p], rec.name[2 * p + 1], score[p]); continue; }
		printf("%s\t%s\tscore=%f\n%s\n%s\n", rec.name[2 * p], rec.name[2 * p + 1], (double)score[p], r1 + stroff[p], r2 + stroff[p]);
	}
	free(blob); free(r1); free(r2); free(off1); free(off2); free(stroff);
	free(l1); free(l2); free(score); free(ei); free(ej); free(st); free(nops);
	at_free_records(&rec);
	free(opt->sites.pos); free(opt);
	return 0;
}

int main(int argc, char *argv[])
{
	int i, ret, cmd = -1, k;
	if (argc < 2) return usage();
	for (k = 0; k < 5; ++k) if (strcmp(argv[1], cmd_name[k]) == 0) cmd = k;
	if (cmd >= 0) ret = main_single(cmd, argc - 1, argv + 1);
	else if (strcmp(argv[1], "batch") == 0) ret = main_batch(argc - 1, argv + 1);
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
