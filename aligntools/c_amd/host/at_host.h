/* at_host.h -- internals of the C host (CLI + reference-shaped wrappers). */
#ifndef AT_HOST_H
#define AT_HOST_H
#include "../../../include/aligntools.h"
#include "../../../include/aligntools_hip.h"

int at_parse_sites(const char *comment, int **pos_out);
/* malloc / realloc / strdup that die() with the reference's mycalloc message (alignment.h:81-87) instead of returning NULL */
void *at_xmalloc(size_t n);
void *at_xrealloc(void *p, size_t n);
char *at_xstrdup(const char *s);
/* ---- incremental block-wise gz FASTA / FASTQ reader (fasta.c) ---- */
typedef struct at_reader at_reader;
typedef struct {
	size_t n, cap;                                      /* records */
	unsigned char *blob; size_t blob_len, blob_cap;     /* the sequences back to back, each followed by a 0 byte */
	size_t *off, *len;                                  /* per record: where its sequence starts in blob, its length */
	char *names; size_t names_len, names_cap; size_t *name_off;            /* 0-terminated names back to back */
	char *comments; size_t comments_len, comments_cap; size_t *comment_off; /* (size_t)-1: the record has no comment */
} at_chunk;
at_reader *at_reader_open(const char *fname);          /* NULL if the file cannot be opened */
size_t at_reader_read(at_reader *r, size_t max_records, size_t max_bases, at_chunk *c);   /* appends to c; 0 = end of file */
void at_reader_close(at_reader *r);
void at_chunk_reset(at_chunk *c);                      /* forget the records, keep the memory */
void at_chunk_free(at_chunk *c);
at_handle *at_host_handle(void);   /* process-wide handle, created on first use; dies without a GPU */
int at_host_handle_exists(void);   /* has this process created it? */

#endif
