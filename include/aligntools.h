/*
 * aligntools.h -- the reference's single-pair function surface, served by the
 * MI355X shim (include/aligntools_hip.h).
 *
 * Same names, argument order, ownership and error behaviour as the five
 * kernels of r3fang/alignTools src/alignment.h, so that the reference's
 * drivers (main_global_affine :476, main_local_affine :851,
 * main_fit_affine_jump :698, main_overlap :967, main_edit_dist :318) compile
 * against this header unchanged:
 *
 *   double align_gla            (s1, s2, r1, r2, opt)   alignment.h:417
 *   double align_local_affine   (s1, s2, r1, r2, opt)   alignment.h:805
 *   double align_fit_affine_jump(s1, s2, r1, r2, opt)   alignment.h:596
 *   double align_overlap        (s1, s2, r1, r2, opt)   alignment.h:926
 *   int    edit_dist            (s1, s2, opt)           alignment.h:291
 *
 * Ownership (alignment.h:176-183, 408-411): the caller owns s1, s2, opt and
 * pre-allocates r1->s / r2->s; the callee frees those two buffers and replaces
 * them with freshly allocated NUL-terminated strings, setting r->l.
 * Errors: NULL arguments, fit with l1 > l2, or a GPU failure end the process
 * through die() -- "FATAL ERROR: <msg>\n" on stderr, exit status 255
 * (alignment.h:69-79).  There is no CPU fallback.
 *
 * Types are layout-compatible with the reference's (kstring.h:56-59,
 * alignment.h:24,51-65).  NB the reference declares
 * `typedef enum { true, false } bool;` -- TRUE IS 0.  opt->s uses that
 * encoding here too (AT_TRUE = 0 means "jump state on").
 */
#ifndef ALIGNTOOLS_H
#define ALIGNTOOLS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef KSTRING_T
#define KSTRING_T kstring_t
typedef struct __kstring_t {
	size_t l, m;
	char *s;
} kstring_t;
#endif

typedef enum { AT_TRUE = 0, AT_FALSE = 1 } at_bool;   /* the reference's inverted enum, alignment.h:24 */

typedef struct {
	size_t size;
	int *pos;
} junction_t;

typedef struct {
	int o;       /* gap open        [-5]  */
	int e;       /* gap extension   [-1]  */
	int m;       /* match           [1]   */
	int u;       /* mismatch        [-2]  */
	int j;       /* jump penalty    [-10] */
	at_bool s;   /* AT_TRUE (0): fit uses the jump state */
	junction_t sites;
} opt_t;

opt_t *init_opt(void);                                  /* alignment.h:102-114 */
void kstring_destory(kstring_t *ks);                    /* alignment.h:206-210 (sic) */
void die(const char *format, ...);                      /* alignment.h:69-79 */

double align_gla(kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, opt_t *opt);
double align_local_affine(kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, opt_t *opt);
double align_fit_affine_jump(kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, opt_t *opt);
double align_overlap(kstring_t *s1, kstring_t *s2, kstring_t *r1, kstring_t *r2, opt_t *opt);
int edit_dist(kstring_t *s1, kstring_t *s2, opt_t *opt);

/*
 * The traceback half of the surface: trace_back_gla :372, trace_back_fit_affine_jump :558, trace_back_local_affine :766,
 * trace_back_overlap :896 -- same names and argument order.  In the reference they walk the eight host matrices of a
 * matrix_t that only the fill loops inside align_*() can produce.  Here the pointer matrix never leaves the GPU and the
 * walk runs in the kernel that filled it, so matrix_t is opaque: it is what a fill left behind -- end cell, start state
 * and the walk's ops -- and comes from at_fill_matrix(), the fill half of align_*() as a call of its own:
 *
 *     matrix_t *S = at_fill_matrix(AT_FILL_LOCAL, s1, s2, opt, &score, &state, &i, &j);   // create_matrix + fill + end cell
 *     trace_back_local_affine(S, s1, s2, r1, r2, i, j);                                   // r1 / r2 as align_*() leaves them
 *     destory_matrix(S);                                                                  // alignment.h:153 (sic)
 *
 * The walk exists for the fill's own end cell only: a trace_back_*() call with another (state, i, j) dies ("FATAL ERROR",
 * rc 255) instead of walking a matrix that is not there.  align_*() = these three calls.
 */
typedef struct at_matrix matrix_t;
enum { AT_FILL_GLOBAL = 0, AT_FILL_LOCAL = 1, AT_FILL_FIT = 2, AT_FILL_OVERLAP = 3 };
/* the reference's state codes (alignment.h:27-34), as trace_back_gla / trace_back_fit_affine_jump take them */
enum { AT_LOW = 500, AT_MID = 600, AT_UPP = 700 };
matrix_t *at_fill_matrix(int fill_mode, kstring_t *s1, kstring_t *s2, opt_t *opt, double *score, int *state, int *i, int *j);
void destory_matrix(matrix_t *S);
void trace_back_gla(matrix_t *S, kstring_t *s1, kstring_t *s2, kstring_t *res_ks1, kstring_t *res_ks2, int state);
void trace_back_fit_affine_jump(matrix_t *S, kstring_t *s1, kstring_t *s2, kstring_t *res_ks1, kstring_t *res_ks2, int state, int i, int j);
void trace_back_local_affine(matrix_t *S, kstring_t *s1, kstring_t *s2, kstring_t *res_ks1, kstring_t *res_ks2, int i, int j);
void trace_back_overlap(matrix_t *S, kstring_t *ks1, kstring_t *ks2, kstring_t *res_ks1, kstring_t *res_ks2, int i, int j);

/* kstring_read (alignment.h:217-262): exactly two FASTA/FASTQ records (plain or
 * gzip) into str1/str2; with opt->s == AT_TRUE also the junction sites from the
 * record comment, echoing the comment to stdout like the reference (:249). */
void kstring_read(char *fname, kstring_t *str1, kstring_t *str2, opt_t *opt);

/* ---- batch extension (new; the reference handles one pair per process) ---- */
typedef struct {
	size_t n;          /* records */
	char **name, **comment, **seq;
	size_t *len;
} at_records;
int at_read_records(const char *fname, at_records *out);   /* 0, or -1 if the file cannot be opened */
void at_free_records(at_records *r);

/* set AT_QUIET_FIT=1 in the environment, or call this with 0, to drop the
 * reference's stray debug line on stdout (alignment.h:602) */
void at_set_fit_debug_line(int on);

#ifdef __cplusplus
}
#endif
#endif
