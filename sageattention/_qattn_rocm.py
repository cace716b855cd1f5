"""``sageattention._qattn_rocm`` (reference pybind module, csrc/qattn/rocm/pybind_gfx942.cpp; same entry names as
``_qattn_sm80`` / ``_qattn_sm89``) -> ctypes shim over libsageattn_hip.so."""
from sageattention_amd._qattn import (qk_int8_sv_f16_accum_f32_attn, qk_int8_sv_f16_accum_f16_attn,  # noqa: F401
                                      qk_int8_sv_f16_accum_f16_attn_inst_buf,
                                      qk_int8_sv_f16_accum_f16_fuse_v_mean_attn,
                                      qk_int8_sv_f8_accum_f32_attn, qk_int8_sv_f8_accum_f32_fuse_v_scale_attn,
                                      qk_int8_sv_f8_accum_f32_fuse_v_scale_attn_inst_buf,
                                      qk_int8_sv_f8_accum_f16_fuse_v_scale_attn_inst_buf,
                                      qk_int8_sv_f8_accum_f32_fuse_v_scale_fuse_v_mean_attn)
