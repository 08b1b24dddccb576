// fp32-compute instantiations of the MFMA GEMM family (v_mfma_f32_16x16x4_f32: exact fp32, the parity mode and
// the loss head's similarity / distance products).
#include <stdlib.h>
#include <type_traits>
#include "gemm_ring.hpp"
#include "gemm_dw_ring.hpp"
#define GEMM_CT float
#define GEMM_TAIL_FN focal_launch_dw_tail_f32
#define GEMM_FN focal_launch_gemm_f32
#include "gemm_dispatch.inc"
