/*
 * abi_consumer.c -- a plain C99 caller of the drop-in boundary (include/aligntools_hip.h and the
 * reference-named surface include/aligntools.h), built by tests/test_c_consumer.py with
 * -std=c99 -pedantic -Wall -Wextra -Werror.
 *
 *   abi_consumer nogpu   host-only entry points; at_init must fail loudly (no CPU fallback)
 *   abi_consumer gpu     the reference's C1 case (test/test_local.fa: local -m 2 -u -2 -o -5 -e -2 -> 4, LEA / MEA)
 *                        through at_align_batch + at_render, at_align_batch_strings, align_local_affine and the fill /
 *                        trace_back_*() pair of calls
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aligntools_hip.h"
#include "aligntools.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static int host_only(void)
{
	/* pack two pairs, 2-bit and 8-bit */
	const uint8_t blob[] = "ACGTACGTTTGACAXYZAC";
	const int64_t off1[2] = {0, 12}, off2[2] = {8, 15};
	const int32_t len1[2] = {8, 3}, len2[2] = {4, 4};
	int bits = -1;
	uint32_t words[32];
	int64_t w1[2], w2[2];
	char r1[8], r2[8];
	const uint8_t ops[3] = {AT_OP_MID, AT_OP_MID, AT_OP_MID};
	at_handle *h = NULL;
	int rc;
	CHECK(at_pack_batch(1, blob, off1, len1, off2, len2, 0, &bits, NULL, w1, w2) == AT_OK && bits == 2);
	CHECK(at_pack_words(1, len1, len2, 2) >= 2);
	CHECK(at_pack_batch(1, blob, off1, len1, off2, len2, 2, NULL, words, w1, w2) == AT_OK);
	CHECK((words[w1[0]] & 0xffffu) == 0xe4e4u);              /* ACGTACGT -> 0,1,2,3,0,1,2,3 */
	CHECK(at_pack_batch(2, blob, off1, len1, off2, len2, 0, &bits, NULL, w1, w2) == AT_OK && bits == 8);
	CHECK(at_pack_batch(2, blob, off1, len1, off2, len2, 2, NULL, words, w1, w2) == AT_ERR_ARG);   /* XYZ has no 2-bit code */
	/* render: 3 diagonal steps ending at (6, 3) of LKSLEA / MEA */
	CHECK(at_render(ops, 3, (const uint8_t *)"LKSLEA", 6, (const uint8_t *)"MEA", 3, r1, r2) == AT_OK);
	CHECK(strcmp(r1, "LEA") == 0 && strcmp(r2, "MEA") == 0);
	rc = at_init(NULL, 0, &h);
	if (rc == AT_OK) { at_destroy(h); printf("host-only ok (a GPU is present)\n"); return 0; }
	CHECK(rc == AT_ERR_NODEVICE && h == NULL);
	CHECK(strstr(at_last_error(NULL), "no CPU fallback") != NULL);
	printf("host-only ok: %s\n", at_last_error(NULL));
	return 0;
}

static int with_gpu(void)
{
	const uint8_t blob[] = "LKSLEAAAAMEAGGG";            /* pair 0: LKSLEA vs MEA (the reference fixture, protein -> byte kernels) */
	const int64_t off1[2] = {0, 6}, off2[2] = {9, 12};     /* pair 1: AAA vs GGG */
	const int32_t len1[2] = {6, 3}, len2[2] = {3, 3};
	const int64_t slot[2] = {0, 16};
	int32_t score[2], ei[2], ej[2], st[2], nops[2];
	uint8_t ops[64];
	char r1[64], r2[64], a[16], b[16];
	at_handle *h = NULL;
	CHECK(at_init(NULL, 0, &h) == AT_OK);
	CHECK(at_set_scoring(h, 2, -2, -5, -2, -10, 0, NULL, 0) == AT_OK);
	CHECK(at_align_batch(h, AT_MODE_LOCAL, 2, blob, off1, len1, off2, len2, 1, score, ei, ej, st, ops, slot, nops) == AT_OK);
	CHECK(score[0] == 4 && ei[0] == 6 && ej[0] == 3 && st[0] == AT_ST_MID && nops[0] == 3);
	CHECK(score[1] == 0);
	CHECK(at_render(ops + slot[0], nops[0], blob + off1[0], ei[0], blob + off2[0], ej[0], a, b) == AT_OK);
	CHECK(strcmp(a, "LEA") == 0 && strcmp(b, "MEA") == 0);
	memset(r1, 'x', sizeof r1); memset(r2, 'x', sizeof r2);
	CHECK(at_align_batch_strings(h, AT_MODE_LOCAL, 2, blob, off1, len1, off2, len2, score, ei, ej, st, r1, r2, slot, nops) == AT_OK);
	CHECK(nops[0] == 3 && strcmp(r1 + slot[0], "LEA") == 0 && strcmp(r2 + slot[0], "MEA") == 0);
	CHECK((int)strlen(r1 + slot[1]) == nops[1] && (int)strlen(r2 + slot[1]) == nops[1]);
	/* errors are codes, not exits */
	CHECK(at_align_batch(h, 99, 1, blob, off1, len1, off2, len2, 0, score, NULL, NULL, NULL, NULL, NULL, NULL) == AT_ERR_ARG);
	CHECK(at_align_batch(h, AT_MODE_FIT, 1, blob, off1, len1, off2, len2, 0, score, NULL, NULL, NULL, NULL, NULL, NULL) == AT_ERR_FIT_ORDER);
	CHECK(strstr(at_last_error(h), "shorter") != NULL);
	at_destroy(h);
	{   /* the reference-named surface: same call shape as main_local_affine (alignment.h:851-890) */
		kstring_t *s1 = (kstring_t *)calloc(1, sizeof *s1), *s2 = (kstring_t *)calloc(1, sizeof *s2);
		kstring_t *q1 = (kstring_t *)calloc(1, sizeof *q1), *q2 = (kstring_t *)calloc(1, sizeof *q2);
		opt_t *opt = init_opt();
		double sc;
		opt->m = 2; opt->u = -2; opt->o = -5; opt->e = -2;
		s1->s = (char *)malloc(8); strcpy(s1->s, "LKSLEA"); s1->l = 6;
		s2->s = (char *)malloc(8); strcpy(s2->s, "MEA"); s2->l = 3;
		q1->s = (char *)calloc(s1->l + s2->l + 1, 1); q2->s = (char *)calloc(s1->l + s2->l + 1, 1);
		sc = align_local_affine(s1, s2, q1, q2, opt);
		CHECK(sc == 4.0 && strcmp(q1->s, "LEA") == 0 && strcmp(q2->s, "MEA") == 0 && q1->l == 3);
		{   /* the fill and the traceback as two calls, the way align_local_affine is written (alignment.h:814-845) */
			double sc2; int state = 0, i = 0, j = 0;
			matrix_t *S = at_fill_matrix(AT_FILL_LOCAL, s1, s2, opt, &sc2, &state, &i, &j);
			CHECK(S != NULL && sc2 == 4.0 && state == AT_MID && i == 6 && j == 3);
			free(q1->s); free(q2->s);
			q1->s = (char *)calloc(s1->l + s2->l + 1, 1); q2->s = (char *)calloc(s1->l + s2->l + 1, 1);
			trace_back_local_affine(S, s1, s2, q1, q2, i, j);
			CHECK(strcmp(q1->s, "LEA") == 0 && strcmp(q2->s, "MEA") == 0 && q2->l == 3);
			destory_matrix(S);
			S = at_fill_matrix(AT_FILL_GLOBAL, s1, s2, opt, &sc2, &state, &i, &j);
			trace_back_gla(S, s1, s2, q1, q2, state);
			CHECK(q1->l == 6 && q2->l == 6 && i == 6 && j == 3);
			destory_matrix(S);
			S = at_fill_matrix(AT_FILL_OVERLAP, s1, s2, opt, &sc2, &state, &i, &j);
			trace_back_overlap(S, s1, s2, q1, q2, i, j);
			CHECK(q1->l == q2->l);
			destory_matrix(S);
			{   /* fit wants l1 <= l2 */
				matrix_t *F = at_fill_matrix(AT_FILL_FIT, s2, s1, opt, &sc2, &state, &i, &j);
				trace_back_fit_affine_jump(F, s2, s1, q1, q2, state, i, j);
				CHECK(q1->l == q2->l && q1->l >= 3 && i == 3);
				destory_matrix(F);
			}
		}
		opt->u = 1;                                           /* -u is the mismatch COST of edit (alignment.h:294) */
		CHECK(edit_dist(s1, s2, opt) == 4);
		free(s1->s); free(s2->s); free(q1->s); free(q2->s); free(s1); free(s2); free(q1); free(q2); free(opt);
	}
	printf("gpu ok\n");
	return 0;
}

int main(int argc, char **argv)
{
	if (argc == 2 && strcmp(argv[1], "nogpu") == 0) return host_only();
	if (argc == 2 && strcmp(argv[1], "gpu") == 0) return with_gpu();
	fprintf(stderr, "usage: abi_consumer nogpu|gpu\n");
	return 2;
}
