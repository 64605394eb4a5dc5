/*
 * ref_reader.c -- TEST INFRASTRUCTURE: a character-at-a-time restatement of the record semantics of the reference's input
 * path (kstring_read, alignment.h:217-262, over klib's kseq_read, kseq.h:189-229).  It was the product's reader in rounds 1
 * and 2; the product now reads block-wise (aligntools/c_amd/host/fasta.c) and tests/test_reader.py checks that reader
 * against this one, record by record, on hostile and randomised inputs.  Not linked into any product binary.
 *
 *   ref_reader <file> ...    prints every record of every file through both readers' common dump format
 *   (built twice by tests/test_reader.py: with -DUSE_PRODUCT_READER against host/fasta.c, and without)
 */
#define _POSIX_C_SOURCE 200809L
#include <ctype.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "aligntools.h"

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

#ifndef USE_PRODUCT_READER
static void *at_xmalloc(size_t n) { void *p = malloc(n ? n : 1); if (!p) die("out of memory"); return p; }
static void *at_xrealloc(void *q, size_t n) { void *p = realloc(q, n ? n : 1); if (!p) die("out of memory"); return p; }
static char *at_xstrdup(const char *s) { size_t n = strlen(s) + 1; char *p = (char *)at_xmalloc(n); memcpy(p, s, n); return p; }

typedef struct {
	gzFile f;
	unsigned char buf[16384];
	int begin, end, eof;
} stream;

static int sgetc(stream *s)
{
	if (s->eof && s->begin >= s->end) return -1;
	if (s->begin >= s->end) {
		s->begin = 0;
		s->end = gzread(s->f, s->buf, sizeof s->buf);
		if (s->end <= 0) { s->end = 0; s->eof = 1; return -1; }
	}
	return s->buf[s->begin++];
}

typedef struct {
	char *s;
	size_t l, m;
} sbuf;

static void sput(sbuf *b, int c)
{
	if (b->l + 2 > b->m) {
		b->m = b->m ? b->m * 2 : 256;
		b->s = (char *)at_xrealloc(b->s, b->m);
	}
	b->s[b->l++] = (char)c;
	b->s[b->l] = 0;
}

/* read up to '\n' (line == 1) or any whitespace (line == 0); returns the delimiter or -1 at EOF,
 * *got = whether anything (even an empty field) was consumed */
static int sgetuntil(stream *s, int line, sbuf *b, int append, int *got)
{
	int c;
	*got = 0;
	if (!append) b->l = 0;
	for (;;) {
		c = sgetc(s);
		if (c < 0) break;
		*got = 1;
		if (line ? c == '\n' : isspace(c)) break;
		sput(b, c);
	}
	/* nothing read and the file is over: ks_getuntil2 returns -1 BEFORE it terminates the string (kseq.h:98-99, 141) -- the length
	 * is 0 but the text, which is what kstring_read looks at (alignment.h:236), is the previous record's, or none */
	if (!*got && c < 0) return c;
	if (!b->s) sput(b, 0), b->l = 0;
	b->s[b->l] = 0;
	if (line && b->l > 1 && b->s[b->l - 1] == '\r') b->s[--b->l] = 0;
	return c;
}

static int ref_read_records(const char *fname, at_records *out)
{
	stream *st;
	sbuf name = {0, 0, 0}, comment = {0, 0, 0}, seq = {0, 0, 0}, qual = {0, 0, 0};
	int c, last = 0, got;
	size_t cap = 0;
	memset(out, 0, sizeof *out);
	st = (stream *)calloc(1, sizeof *st);
	if (!st) return -1;
	st->f = gzopen(fname, "r");
	if (!st->f) { free(st); return -1; }
	for (;;) {
		if (last == 0) {
			while ((c = sgetc(st)) != -1 && c != '>' && c != '@') {}
			if (c == -1) break;
			last = c;
		}
		seq.l = 0;
		c = sgetuntil(st, 0, &name, 0, &got);
		if (!got && c < 0) break;
		if (c != '\n' && c >= 0) sgetuntil(st, 1, &comment, 0, &got);
		while ((c = sgetc(st)) != -1 && c != '>' && c != '+' && c != '@') {
			if (c == '\n') continue;
			sput(&seq, c);
			sgetuntil(st, 1, &seq, 1, &got);
		}
		last = (c == '>' || c == '@') ? c : 0;
		if (c == '+') {
			while ((c = sgetc(st)) != -1 && c != '\n') {}
			if (c == -1) break;                     /* no quality string: kseq_read returns -2 */
			qual.l = 0;
			for (;;) {
				int d = sgetuntil(st, 1, &qual, 1, &got);
				if ((!got && d < 0) || qual.l >= seq.l) break;
			}
			last = 0;
			if (qual.l != seq.l) break;            /* truncated quality: -2 ends the reader loop */
		}
		if (out->n == cap) {
			cap = cap ? cap * 2 : 4;
			out->name = (char **)at_xrealloc(out->name, cap * sizeof(char *));
			out->comment = (char **)at_xrealloc(out->comment, cap * sizeof(char *));
			out->seq = (char **)at_xrealloc(out->seq, cap * sizeof(char *));
			out->len = (size_t *)at_xrealloc(out->len, cap * sizeof(size_t));
		}
		out->name[out->n] = at_xstrdup(name.s ? name.s : "");
		out->comment[out->n] = comment.s ? at_xstrdup(comment.s) : NULL;   /* the shared-buffer quirk */
		out->seq[out->n] = (char *)at_xmalloc(seq.l + 1);
		memcpy(out->seq[out->n], seq.s ? seq.s : "", seq.l);
		out->seq[out->n][seq.l] = 0;
		out->len[out->n] = seq.l;
		out->n++;
	}
	free(name.s); free(comment.s); free(seq.s); free(qual.s);
	gzclose(st->f);
	free(st);
	return 0;
}

#define READ ref_read_records
#else
#define READ at_read_records
#endif

static void dump_bytes(const char *s, size_t n)
{
	size_t k;
	for (k = 0; k < n; ++k) {
		const unsigned char ch = (unsigned char)s[k];
		if (ch > 32 && ch < 127 && ch != '\\') putchar(ch); else printf("\\x%02x", ch);
	}
}

int main(int argc, char **argv)
{
	int k;
	for (k = 1; k < argc; ++k) {
		at_records rec;
		size_t r;
		if (READ(argv[k], &rec) != 0) { printf("%s: cannot open\n", argv[k]); continue; }
		printf("%s: %d records\n", argv[k], (int)rec.n);
		for (r = 0; r < rec.n; ++r) {
			printf("  name=["); dump_bytes(rec.name[r], strlen(rec.name[r]));
			printf("] comment="); if (rec.comment[r]) { putchar('['); dump_bytes(rec.comment[r], strlen(rec.comment[r])); putchar(']'); } else printf("NULL");
			printf(" len=%d seq=[", (int)rec.len[r]); dump_bytes(rec.seq[r], rec.len[r]); printf("]\n");
		}
	}
	return 0;
}
