// bf16-compute instantiations of the MFMA GEMM family (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
#include <stdlib.h>
#include <type_traits>
#include "gemm_ring.hpp"
#include "gemm_dw_ring.hpp"
#define GEMM_CT bf16_t
#define GEMM_TAIL_FN focal_launch_dw_tail_bf16
#define GEMM_FN focal_launch_gemm_bf16
#include "gemm_dispatch.inc"
