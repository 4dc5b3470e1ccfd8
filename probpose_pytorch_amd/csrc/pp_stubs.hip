// TEMPORARY: entry points not implemented yet (removed as the kernels land).
#include "pp_common.h"
extern "C" int pp_gemm(const pp_gemm_args *, void *) { return pp::fail("pp_gemm: not built yet"); }
extern "C" int pp_layernorm(const float *, const float *, const float *, float, int, int, void *, int, void *) { return pp::fail("not built yet"); }
extern "C" int pp_attention(const void *, void *, int, int, int, int, int, void *) { return pp::fail("not built yet"); }
extern "C" int pp_patchify(const float *, void *, int, int, int, int, int, void *) { return pp::fail("not built yet"); }
extern "C" int pp_maxpool_relu(const void *, void *, int, int, int, int, int, int, int, void *) { return pp::fail("not built yet"); }
extern "C" int pp_final_heatmap(const void *, const void *, const float *, float *, int, int, int, int, float, int, void *) { return pp::fail("not built yet"); }
extern "C" int pp_aux_tail(const void *, const void *, const float *, float *, int, int, int, int, void *) { return pp::fail("not built yet"); }
extern "C" int pp_tokens_to_nchw(const void *, float *, int, int, int, int, void *) { return pp::fail("not built yet"); }
extern "C" int pp_nchw_to_tokens(const float *, void *, int, int, int, int, void *) { return pp::fail("not built yet"); }
