// Entry points declared in include/dvg.h whose kernels are not written yet.
// They fail loudly (DVG_E_UNSUPPORTED); nothing falls back to another path.
#include "common.h"
using namespace dvg;
#define NOT_YET(name) do { set_error(name ": not implemented in this build"); return DVG_E_UNSUPPORTED; } while (0)
extern "C" {
size_t dvg_encoder_workspace_bytes(int64_t, int) { return 0; }
int dvg_encoder_fwd(const dvg_encoder_params_t*, int, const float*, int64_t, int, float*, void*, size_t, dvg_stream_t) { NOT_YET("dvg_encoder_fwd"); }
int dvg_encoder_bwd(const dvg_encoder_params_t*, int, const float*, int64_t, const float*, const dvg_encoder_grads_t*, void*, size_t, dvg_stream_t) { NOT_YET("dvg_encoder_bwd"); }
size_t dvg_decoder_workspace_bytes(int64_t, int) { return 0; }
int dvg_decoder_fwd(const dvg_decoder_params_t*, int, const float*, int64_t, int, const float* const*, uint64_t, uint64_t, float*, void*, size_t, dvg_stream_t) { NOT_YET("dvg_decoder_fwd"); }
int dvg_decoder_bwd(const dvg_decoder_params_t*, int, const float*, int64_t, const float*, const dvg_decoder_grads_t*, float*, void*, size_t, dvg_stream_t) { NOT_YET("dvg_decoder_bwd"); }
}
