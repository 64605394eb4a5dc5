/*
 * at_oracle.h -- TEST INFRASTRUCTURE ONLY (the parity checker).
 *
 * CPU restatement of the five dynamic-programming kernels of
 * r3fang/alignTools (reference src/alignment.h).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may link or load this; the
 * product (aligntools/c_amd) never does.
 *
 * Parity pinning: the reference ships NO golden outputs of its own
 * (SURVEY.md section 4), so this restatement is pinned against the real
 * reference compiled in place (oracle/_ref/libat_ref.so, see oracle/Makefile)
 * on (a) every known answer in SURVEY.md section 4 and (b) the randomized
 * fixtures committed under tests/golden/ by oracle/make_golden.py.
 */
#ifndef AT_ORACLE_H
#define AT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* same numbering as include/aligntools_hip.h */
enum { ATO_GLOBAL = 0, ATO_LOCAL = 1, ATO_FIT = 2, ATO_OVERLAP = 3, ATO_EDIT = 4 };

/* traceback op codes, emitted end -> start (traceback order) */
enum { ATO_OP_MID = 0,  /* (s1[i-1], s2[j-1])  i--, j--                         */
       ATO_OP_LOW = 1,  /* (s1[i-1], '-')      i--                               */
       ATO_OP_UPP = 2,  /* ('-', s2[j-1])      j--                               */
       ATO_OP_JUMP = 3  /* ('-', s2[j-1])      j--  (fit jump state, rendered as UPP) */ };

/* start states reported by ato_align (state the traceback starts in) */
enum { ATO_ST_LOW = 1, ATO_ST_MID = 2, ATO_ST_UPP = 3 };

typedef struct {
	int m, u, o, e, j;   /* match, mismatch, gap open, gap extension, jump (opt_t, alignment.h:57-65) */
	int use_jump;        /* fit -s */
	const int *sites;    /* junction sites, 0-based positions on s2 */
	int nsites;
} ato_scoring;

/*
 * Align one pair.  Returns 0 on success,
 *   -1 bad argument / capacity, -2 input outside the domain on which the
 *   reference is defined (it would read uninitialised memory or loop forever).
 * r1/r2: the two gapped strings, NUL terminated (cap >= l1+l2+1).
 * ops:   optional (may be NULL) op codes in traceback order, *nops of them.
 * end_i/end_j: cell the traceback starts from; start_state: ATO_ST_*.
 */
int ato_align(int mode, const char *s1, int l1, const char *s2, int l2,
              const ato_scoring *sc, double *score,
              char *r1, char *r2, int cap, int *rlen,
              int *end_i, int *end_j, int *start_state,
              unsigned char *ops, int *nops);

/* timed loop over n fixed-shape pairs laid out s1,s2,s1,s2,... in blob */
double ato_time_batch(int mode, int n, const char *blob, int l1, int l2,
                      const ato_scoring *sc, double *checksum);

#ifdef __cplusplus
}
#endif
#endif
