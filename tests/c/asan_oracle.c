/*
 * asan_oracle.c -- the oracle restatement (oracle/at_oracle.c, test infrastructure) under AddressSanitizer + UBSan:
 * every mode on seeded random pairs over the length edges the GPU tiling cares about (1, 2, 63..65, 127..129, 150),
 * with and without the jump state.  Prints a checksum of the scores; the sanitizers do the checking.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "at_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static unsigned rnd(void)
{
	rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
	return (unsigned)((rng_state * 0x2545F4914F6CDD1Dull) >> 33);
}

int main(void)
{
	static const int edges[] = {1, 2, 3, 17, 63, 64, 65, 127, 128, 129, 150};
	const int ne = (int)(sizeof edges / sizeof edges[0]);
	long long sum = 0;
	int mode, a, b, rep;
	for (mode = 0; mode <= 4; ++mode) {
		for (a = 0; a < ne; ++a) {
			for (b = 0; b < ne; b += 2) {
				for (rep = 0; rep < 2; ++rep) {
					int l1 = edges[a], l2 = edges[b] + (int)(rnd() % 40), k, nsites = 0, sites[4];
					char *s1, *s2, *r1, *r2;
					unsigned char *ops;
					ato_scoring sc;
					double score = 0;
					int rlen = 0, ei = 0, ej = 0, st = 0, nops = 0, rc;
					if (mode == ATO_FIT && l1 > l2) { int t = l1; l1 = l2; l2 = t; }
					if (mode == ATO_FIT && l2 < 2) l2 = 2;
					s1 = (char *)malloc((size_t)l1 + 1); s2 = (char *)malloc((size_t)l2 + 1);
					r1 = (char *)malloc((size_t)l1 + l2 + 1); r2 = (char *)malloc((size_t)l1 + l2 + 1);
					ops = (unsigned char *)malloc((size_t)l1 + l2 + 1);
					for (k = 0; k < l1; ++k) s1[k] = "ACGT"[rnd() & 3];
					for (k = 0; k < l2; ++k) s2[k] = rep ? s1[k % l1] : "ACGT"[rnd() & 3];
					s1[l1] = 0; s2[l2] = 0;
					if (mode == ATO_FIT && rep) { nsites = 3; sites[0] = l2 / 4; sites[1] = l2 / 2; sites[2] = l2 + 5; }
					sc.m = 2; sc.u = -2; sc.o = -5; sc.e = -1; sc.j = -10; sc.use_jump = nsites > 0; sc.sites = sites; sc.nsites = nsites;
					rc = ato_align(mode, s1, l1, s2, l2, &sc, &score, r1, r2, l1 + l2 + 1, &rlen, &ei, &ej, &st, ops, &nops);
					if (rc != 0) { fprintf(stderr, "oracle failed (%d): mode %d %dx%d\n", rc, mode, l1, l2); return 1; }
					if (mode != ATO_EDIT && (int)strlen(r1) != rlen) { fprintf(stderr, "string length: mode %d\n", mode); return 1; }
					sum += (long long)score + nops;
					free(r1); free(r2);
					free(s1); free(s2); free(ops);
				}
			}
		}
	}
	printf("asan_oracle ok: checksum %lld\n", sum);
	return 0;
}
