/*
 * kseq_harness.c -- TEST INFRASTRUCTURE, build container only: the reference's own reader (klib's kseq.h, found through
 * -I/root/reference/src; nothing of it is copied here) behind the dump format of ref_reader.c, so that tests/test_reader.py can hold
 * the product's block-wise reader (host/fasta.c) and the character-at-a-time restatement (ref_reader.c) against the real thing.
 * The comment is reported the way kstring_read sees it (alignment.h:236: the C string in seq->comment.s, whatever its length field says).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <zlib.h>
#include "kseq.h"

KSEQ_INIT(gzFile, gzread)

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
		gzFile fp = gzopen(argv[k], "r");
		kseq_t *seq;
		int n = 0;
		if (!fp) { printf("%s: cannot open\n", argv[k]); continue; }
		seq = kseq_init(fp);
		/* two passes: the record count first (the dump format leads with it) */
		while (kseq_read(seq) >= 0) ++n;
		kseq_destroy(seq); gzclose(fp);
		printf("%s: %d records\n", argv[k], n);
		fp = gzopen(argv[k], "r");
		seq = kseq_init(fp);
		while (kseq_read(seq) >= 0) {
			printf("  name=["); dump_bytes(seq->name.s ? seq->name.s : "", seq->name.s ? strlen(seq->name.s) : 0);
			printf("] comment="); if (seq->comment.s) { putchar('['); dump_bytes(seq->comment.s, strlen(seq->comment.s)); putchar(']'); } else printf("NULL");
			printf(" len=%d seq=[", (int)seq->seq.l); dump_bytes(seq->seq.s ? seq->seq.s : "", seq->seq.l); printf("]\n");
		}
		kseq_destroy(seq); gzclose(fp);
	}
	return 0;
}
