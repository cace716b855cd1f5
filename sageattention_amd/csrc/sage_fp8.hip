// FP8 (OCP e4m3fn) V quantizer and INT8-QK / FP8-PV attention -- placeholder entry points.
#include "sage_common.h"
extern "C" size_t sage_quant_v_fp8_workspace_bytes(int, int, int, int) { return 0; }
extern "C" int sage_quant_v_fp8(const sage_tensor*, int, int, int, int, int, const sage_tensor*, float*, float*, float,
                                void*, sage_stream_t) { return SAGE_ERR_UNSUPPORTED; }
extern "C" int sage_attn_qk_int8_pv_f8(const sage_tensor*, const sage_tensor*, const sage_tensor*, const sage_tensor*, int,
                                       const float*, const float*, const float*, const float*, float*, int, int, int, int,
                                       int, int, int, int, int, int, float, int, sage_stream_t) { return SAGE_ERR_UNSUPPORTED; }
