/*
 * fasta.c -- gz FASTA/FASTQ reader of the C host: block-wise, incremental.
 *
 * The reader works on lines found with memchr in a multi-megabyte window over zlib's gzread and appends every
 * sequence straight to the batch's sequence blob (the layout at_align_batch takes), so a batch of 200 000 reads costs
 * one pass over its bytes and no per-record allocation.  at_reader_read() hands out the next records of the file; the
 * batch driver (cli.c) parses a chunk while the GPU aligns the previous one.
 *
 * Record semantics = the reference's input path (kstring_read, alignment.h:217-262, over klib's kseq_read,
 * kseq.h:189-229) -- checked byte for byte against a character-at-a-time restatement (tests/c/ref_reader.c) on hostile
 * inputs and against the stock binary's recorded outputs:
 *   - everything before the first '>' or '@' is skipped (any position, not only a line start);
 *   - header line: name = up to the first whitespace; if that whitespace is not the newline, comment = rest of the line
 *     (a trailing '\r' dropped when more than one character is left);
 *   - sequence = the following lines up to one that STARTS with '>', '@' or '+' (empty lines skipped; a '\r' at the
 *     end of what has been collected so far is dropped after every line, when more than one character is there);
 *   - '+' starts a FASTQ quality block: the rest of its line is skipped, then quality lines are read until they cover
 *     the sequence; a block that ends early or overshoots ends the file without that record;
 *   - kseq keeps ONE comment buffer for the whole file and kstring_read copies it whenever it is non-NULL
 *     (alignment.h:235): a record without a comment inherits the text of the last record that had one.  Reproduced.
 * Plain and gzip input both go through gzread.
 */
#define _POSIX_C_SOURCE 200809L
#include "at_host.h"
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* allocation failures end the process like the reference's mycalloc (alignment.h:81-87) */
void *at_xmalloc(size_t n)
{
	void *p = malloc(n ? n : 1);
	if (!p) die("mycalloc failure requesting %d of size %d bytes", (int)n, 1);
	return p;
}
void *at_xrealloc(void *q, size_t n)
{
	void *p = realloc(q, n ? n : 1);
	if (!p) die("mycalloc failure requesting %d of size %d bytes", (int)n, 1);
	return p;
}
char *at_xstrdup(const char *s)
{
	size_t n = strlen(s) + 1;
	char *p = (char *)at_xmalloc(n);
	memcpy(p, s, n);
	return p;
}

struct at_reader {
	gzFile f;
	unsigned char *buf;
	size_t cap, begin, end;
	int eof;
	int last;                 /* the '>' / '@' that ended the previous record's sequence, 0 = must search */
	char *comment;            /* kseq's one comment buffer: NULL until a header with a comment has been seen */
	size_t comment_len, comment_cap;
	unsigned char *qual;      /* the quality block of the record being read (only its length and '\r' handling matter) */
	size_t qual_cap;
};

#define AT_READER_WINDOW ((size_t)4 << 20)

static int is_space(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }   /* isspace() of the C locale */

at_reader *at_reader_open(const char *fname)
{
	at_reader *r = (at_reader *)calloc(1, sizeof *r);
	if (!r) return NULL;
	r->f = gzopen(fname, "r");
	if (!r->f) { free(r); return NULL; }
	(void)gzbuffer(r->f, 1u << 20);
	r->cap = AT_READER_WINDOW;
	r->buf = (unsigned char *)at_xmalloc(r->cap);
	return r;
}

void at_reader_close(at_reader *r)
{
	if (!r) return;
	gzclose(r->f);
	free(r->buf); free(r->comment); free(r->qual);
	free(r);
}

/* more bytes behind buf[end); the unread part moves to the front first.  0 at the end of the file. */
static int refill(at_reader *r)
{
	int got;
	if (r->eof) return 0;
	if (r->begin > 0) {
		memmove(r->buf, r->buf + r->begin, r->end - r->begin);
		r->end -= r->begin; r->begin = 0;
	}
	if (r->end == r->cap) {            /* one line longer than the window: widen it */
		r->cap *= 2;
		r->buf = (unsigned char *)at_xrealloc(r->buf, r->cap);
	}
	got = gzread(r->f, r->buf + r->end, (unsigned)(r->cap - r->end > ((size_t)1 << 30) ? (size_t)1 << 30 : r->cap - r->end));
	if (got <= 0) { r->eof = 1; return 0; }
	r->end += (size_t)got;
	return 1;
}

static int peekc(at_reader *r)
{
	if (r->begin >= r->end && !refill(r)) return -1;
	return r->buf[r->begin];
}

/* the rest of the current line: *p .. *p + *n (inside the window, valid until the next call), the '\n' consumed.
 * Returns 1 if a '\n' ended it, 0 if the file did (then *n may still be > 0), and sets *any to whether a single byte
 * (the newline included) was consumed at all. */
static int get_line(at_reader *r, const unsigned char **p, size_t *n, int *any)
{
	size_t scanned = 0;
	for (;;) {
		const unsigned char *nl = (const unsigned char *)memchr(r->buf + r->begin + scanned, '\n', r->end - r->begin - scanned);
		if (nl) {
			*p = r->buf + r->begin; *n = (size_t)(nl - *p);
			r->begin += *n + 1;
			*any = 1;
			return 1;
		}
		scanned = r->end - r->begin;
		if (!refill(r)) {
			*p = r->buf + r->begin; *n = r->end - r->begin;
			r->begin = r->end;
			*any = *n > 0;
			return 0;
		}
	}
}

static void grow_bytes(void **p, size_t *cap, size_t need)
{
	if (need <= *cap) return;
	while (*cap < need) *cap = *cap ? *cap * 2 : 4096;
	*p = at_xrealloc(*p, *cap);
}

void at_chunk_reset(at_chunk *c)
{
	c->n = 0; c->blob_len = 0; c->names_len = 0; c->comments_len = 0;
}

void at_chunk_free(at_chunk *c)
{
	free(c->blob); free(c->off); free(c->len); free(c->names); free(c->name_off); free(c->comments); free(c->comment_off);
	memset(c, 0, sizeof *c);
}

static size_t put_text(char **arena, size_t *len, size_t *cap, const void *s, size_t n)
{
	const size_t at = *len;
	grow_bytes((void **)arena, cap, *len + n + 1);
	memcpy(*arena + at, s, n);
	(*arena)[at + n] = 0;
	*len = at + n + 1;
	return at;
}

/* Append up to max_records records (and stop behind the record that takes the chunk's blob past max_bases) to `c`.
 * Returns the number appended; 0 = the file has no more records. */
size_t at_reader_read(at_reader *r, size_t max_records, size_t max_bases, at_chunk *c)
{
	size_t added = 0;
	const unsigned char *p;
	size_t n;
	int any, c0;
	while (added < max_records && c->blob_len < max_bases) {
		size_t name_at, seq_at, seq_len, k;
		if (r->last == 0) {
			/* skip to the next '>' or '@', wherever it is */
			for (;;) {
				const unsigned char *a, *b;
				size_t span;
				if (r->begin >= r->end && !refill(r)) return added;
				span = r->end - r->begin;
				a = (const unsigned char *)memchr(r->buf + r->begin, '>', span);
				b = (const unsigned char *)memchr(r->buf + r->begin, '@', a ? (size_t)(a - (r->buf + r->begin)) : span);
				if (b) a = b;
				if (a) { r->last = *a; r->begin = (size_t)(a - r->buf) + 1; break; }
				r->begin = r->end;
			}
		}
		/* header line */
		{
			const int nl_ended = get_line(r, &p, &n, &any);
			if (!any) return added;                         /* the file ends right behind the marker: no record (kseq -1) */
			for (k = 0; k < n && !is_space(p[k]); ++k) {}
			name_at = put_text(&c->names, &c->names_len, &c->names_cap, p, k);
			/* a whitespace other than the line's newline: the rest is the comment.  When the FILE ends right behind that whitespace
			 * kseq's ks_getuntil(comment) reads nothing, returns -1 and leaves the comment buffer as it was -- the previous
			 * record's text, or none (kseq.h:189-229) */
			if (k < n && (k + 1 < n || nl_ended)) {
				size_t cl = n - k - 1;
				if (cl > 1 && p[n - 1] == '\r') --cl;
				grow_bytes((void **)&r->comment, &r->comment_cap, cl + 1);
				memcpy(r->comment, p + k + 1, cl);
				r->comment[cl] = 0; r->comment_len = cl;
			}
		}
		/* sequence lines */
		seq_at = c->blob_len; seq_len = 0;
		for (;;) {
			c0 = peekc(r);
			if (c0 < 0) { r->last = 0; break; }
			if (c0 == '>' || c0 == '@' || c0 == '+') { ++r->begin; r->last = c0; break; }
			if (c0 == '\n') { ++r->begin; continue; }
			{
				const int nl_ended = get_line(r, &p, &n, &any);
				grow_bytes((void **)&c->blob, &c->blob_cap, seq_at + seq_len + n + 1);
				memcpy(c->blob + seq_at + seq_len, p, n);
				seq_len += n;
				/* kseq takes a line's first byte with ks_getc and the rest with ks_getuntil2, which drops a trailing '\r' -- unless it
				 * read nothing and the file is over (it returns before that, kseq.h:141): a lone last byte '\r' stays */
				if (n + (size_t)nl_ended >= 2 && seq_len > 1 && c->blob[seq_at + seq_len - 1] == '\r') --seq_len;
			}
		}
		if (r->last == '+') {
			size_t ql = 0;
			const int got_nl = get_line(r, &p, &n, &any);    /* the rest of the '+' line */
			r->last = 0;
			if (!got_nl) return added;                        /* no quality string at all: the record is lost (kseq -2) */
			for (;;) {
				(void)get_line(r, &p, &n, &any);
				if (!any) break;
				grow_bytes((void **)&r->qual, &r->qual_cap, ql + n + 1);
				memcpy(r->qual + ql, p, n);
				ql += n;
				if (ql > 1 && r->qual[ql - 1] == '\r') --ql;
				if (ql >= seq_len) break;
			}
			if (ql != seq_len) { r->eof = 1; r->begin = r->end; return added; }   /* truncated or overlong quality: the reader stops here */
		}
		grow_bytes((void **)&c->blob, &c->blob_cap, seq_at + seq_len + 1);
		c->blob[seq_at + seq_len] = 0;
		c->blob_len = seq_at + seq_len + 1;
		if (c->n == c->cap) {
			c->cap = c->cap ? c->cap * 2 : 1024;
			c->off = (size_t *)at_xrealloc(c->off, c->cap * sizeof(size_t));
			c->len = (size_t *)at_xrealloc(c->len, c->cap * sizeof(size_t));
			c->name_off = (size_t *)at_xrealloc(c->name_off, c->cap * sizeof(size_t));
			c->comment_off = (size_t *)at_xrealloc(c->comment_off, c->cap * sizeof(size_t));
		}
		c->off[c->n] = seq_at; c->len[c->n] = seq_len; c->name_off[c->n] = name_at;
		c->comment_off[c->n] = r->comment ? put_text(&c->comments, &c->comments_len, &c->comments_cap, r->comment, r->comment_len) : (size_t)-1;
		++c->n; ++added;
	}
	return added;
}

/* the whole file as separately allocated records (the single-pair drivers and callers of include/aligntools.h) */
int at_read_records(const char *fname, at_records *out)
{
	at_reader *r;
	at_chunk c;
	size_t k;
	memset(out, 0, sizeof *out);
	memset(&c, 0, sizeof c);
	r = at_reader_open(fname);
	if (!r) return -1;
	while (at_reader_read(r, (size_t)-1, (size_t)-1, &c) > 0) {}
	at_reader_close(r);
	out->n = c.n;
	out->name = (char **)at_xmalloc((c.n + 1) * sizeof(char *));
	out->comment = (char **)at_xmalloc((c.n + 1) * sizeof(char *));
	out->seq = (char **)at_xmalloc((c.n + 1) * sizeof(char *));
	out->len = (size_t *)at_xmalloc((c.n + 1) * sizeof(size_t));
	for (k = 0; k < c.n; ++k) {
		out->name[k] = at_xstrdup(c.names + c.name_off[k]);
		out->comment[k] = c.comment_off[k] == (size_t)-1 ? NULL : at_xstrdup(c.comments + c.comment_off[k]);
		out->seq[k] = (char *)at_xmalloc(c.len[k] + 1);
		memcpy(out->seq[k], c.blob + c.off[k], c.len[k] + 1);
		out->len[k] = c.len[k];
	}
	at_chunk_free(&c);
	return 0;
}

void at_free_records(at_records *r)
{
	size_t k;
	for (k = 0; k < r->n; ++k) { free(r->name[k]); free(r->comment[k]); free(r->seq[k]); }
	free(r->name); free(r->comment); free(r->seq); free(r->len);
	memset(r, 0, sizeof *r);
}

/* ksplit(tmp, '|', &n) + atoi per field (alignment.h:250-253, kstring.c:89-131): fields are the
 * maximal runs of non-'|' characters */
int at_parse_sites(const char *comment, int **pos_out)
{
	size_t l = strlen(comment), i, cap = 8;
	int n = 0, *pos = (int *)at_xmalloc(cap * sizeof(int));
	i = 0;
	while (i < l) {
		while (i < l && comment[i] == '|') ++i;
		if (i >= l) break;
		if ((size_t)n == cap) { cap *= 2; pos = (int *)at_xrealloc(pos, cap * sizeof(int)); }
		pos[n++] = atoi(comment + i);
		while (i < l && comment[i] != '|') ++i;
	}
	*pos_out = pos;
	return n;
}
