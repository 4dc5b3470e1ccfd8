// Multi-head self-attention entry point (timm Attention.forward semantics:
// softmax(q k^T * hd^-1/2) v per (crop, head); N = 192 or 432 tokens, so one
// (crop, head) problem fits on a CU and no online-softmax tiling over HBM is
// needed).  Dispatch: bf16 + supported head dim -> MFMA kernel; everything
// else (fp32 parity mode) -> exact-fp32 VALU kernel in pp_ops.hip.
#include "pp_common.h"

namespace pp {
template <typename T>
int attention_valu(const void *qkv, void *out, int B, int N, int heads, int hd, hipStream_t s);
}

extern "C" int pp_attention(const void *qkv, void *out, int B, int N, int heads, int hd, int dtype,
                            void *stream) {
  using namespace pp;
  PP_REQUIRE(B >= 0 && N > 0 && heads > 0 && hd > 0, "pp_attention: bad shape");
  if (B == 0) return 0;
  PP_REQUIRE(qkv && out, "pp_attention: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PP_BF16) return attention_valu<bf16_t>(qkv, out, B, N, heads, hd, s);
  if (dtype == PP_F32) return attention_valu<float>(qkv, out, B, N, heads, hd, s);
  return fail("pp_attention: bad dtype %d", dtype);
}
