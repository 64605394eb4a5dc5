/*
 * asan_parser.c -- the C host's input path under AddressSanitizer + UBSan (CPU only, no HIP in the process).
 *
 *   asan_parser <file> ...     every file goes through at_read_records (gz FASTA / FASTQ reader, host/fasta.c) and every
 *                              record comment through at_parse_sites; everything is freed again, so LeakSanitizer sees the
 *                              reader's own bookkeeping.  Prints one line per file: records, total bases, sites parsed.
 *
 * Built by `make -C aligntools/c_amd asan` from host/fasta.c and this file alone; die() is defined here (the product's
 * lives in host/compat.c next to the GPU calls).
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aligntools.h"

int at_parse_sites(const char *comment, int **pos_out);

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

int main(int argc, char **argv)
{
	int k;
	for (k = 1; k < argc; ++k) {
		at_records rec;
		size_t r, bases = 0;
		long sites = 0;
		if (at_read_records(argv[k], &rec) != 0) { printf("%s: cannot open\n", argv[k]); continue; }
		for (r = 0; r < rec.n; ++r) {
			bases += rec.len[r];
			if (rec.seq[r][rec.len[r]] != 0) { fprintf(stderr, "%s: record %d: not terminated\n", argv[k], (int)r); return 1; }
			if (rec.comment[r]) {
				int *pos = NULL;
				sites += at_parse_sites(rec.comment[r], &pos);
				free(pos);
			}
		}
		printf("%s: %d records, %d bases, %ld sites\n", argv[k], (int)rec.n, (int)bases, sites);
		at_free_records(&rec);
	}
	return 0;
}
