#include "at_launch.h"
/* packed local kernels for RAGGED batches: four groups of 16 lanes sweep a frame of K * 16 rows x l2 columns in which
 * every alignment keeps its own extents (at_sweep16.hip.h, RAG); pairs of similar size are put together by the host */
template <int K>
static at_sweep16_fn r3(int store, bool tb)
{
	if (!tb) return at::at_sweep16<at::K_LOCAL, 16, K, 4, true, true, false, true, AT_BITS16>;
	if (store == 0) return at::at_sweep16<at::K_LOCAL, 16, K, 4, true, true, true, true, AT_BITS16>;
	return at::at_sweep16<at::K_LOCAL, 16, K, 4, true, false, true, true, AT_BITS16>;
}
at_sweep16_fn AT_NAME(at_pick16_rag_impl)(int k, int store, bool tb)
{
	switch (k) {
	case 4: return r3<4>(store, tb);
	case 5: return r3<5>(store, tb);
	case 6: return r3<6>(store, tb);
	case 7: return r3<7>(store, tb);
	case 10: return r3<10>(store, tb);
	case 13: return r3<13>(store, tb);
	default: return nullptr;
	}
}
