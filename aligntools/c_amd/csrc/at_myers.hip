#include "at_myers.hip.h"
#include "at_launch.h"
/* bit-parallel edit distance: W words of 32 rows per lane, G lanes per alignment (l1 <= 32 * G * W: up to 32 768 bases) */
at_myers_fn at_pick_myers(int w, int g)
{
	if (g == 1)   /* one alignment per lane: reads up to 64 / 96 / 128 / 160 / 256 / 512 / 1 024 bases */
		return w == 2 ? at::at_myers<2, 1> : w == 3 ? at::at_myers<3, 1> : w == 4 ? at::at_myers<4, 1> : w == 5 ? at::at_myers<5, 1> : w == 8 ? at::at_myers<8, 1>
		     : w == 16 ? at::at_myers<16, 1> : w == 32 ? at::at_myers<32, 1> : nullptr;
	if (g == 8) return w == 1 ? at::at_myers<1, 8> : nullptr;   /* reads up to 256 bases, 8 alignments per wavefront */
	switch (w) {
	case 1: return at::at_myers<1, 32>;
	case 2: return at::at_myers<2, 32>;
	case 4: return at::at_myers<4, 32>;
	case 8: return at::at_myers<8, 32>;
	case 16: return at::at_myers<16, 32>;   /* reads up to 16 384 / 32 768 bases */
	case 32: return at::at_myers<32, 32>;
	default: return nullptr;
	}
}
/* the overlap filter (at_myers<W, 1, true>): one alignment per lane, reads of up to 32 * W bases */
at_myers_fn at_pick_myers_semi(int w)
{
	return w == 2 ? at::at_myers<2, 1, true> : w == 3 ? at::at_myers<3, 1, true> : w == 4 ? at::at_myers<4, 1, true> : w == 5 ? at::at_myers<5, 1, true>
	     : w == 8 ? at::at_myers<8, 1, true> : w == 16 ? at::at_myers<16, 1, true> : w == 32 ? at::at_myers<32, 1, true> : nullptr;
}
