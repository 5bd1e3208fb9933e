/*
 * oracle/msda_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, gcc) of the reference's multi-scale deformable
 * attention forward/backward.  It is the checker for tests/, for
 * __graft_entry__.smoke() and for bench.py's cpu_baseline leg.  Nothing under
 * vnext_amd/ may import, link or call it: the product path is the HIP library.
 *
 * Parity pin: oracle/make_golden.py imports the reference's own pure-PyTorch
 * function (projects/SeqFormer/seqformer/models/ops/functions/
 * ms_deform_attn_func.py:42-62) in the build container and stores its outputs
 * and autograd gradients under tests/golden/; tests/test_oracle.py checks this
 * file against every one of those vectors.
 *
 * What each routine follows (paths relative to
 * projects/SeqFormer/seqformer/models/ops/src/cuda/):
 *   sample geometry  ms_deform_im2col_cuda.cuh:285-288  (x*W-0.5, y*H-0.5, open
 *                    interval test h>-1 && w>-1 && h<H && w<W)
 *   corner rules     ms_deform_im2col_cuda.cuh:38-82    (floor, 4 guarded taps)
 *   forward sum      ms_deform_im2col_cuda.cuh:253-298
 *   backward         ms_deform_im2col_cuda.cuh:87-159 (per-tap gradients) and
 *                    :301-403 (reduction over channels, writes of the
 *                    location / weight gradients)
 *   size derivation  ms_deform_attn_cuda.cu:40-60
 *
 * Layouts (all contiguous, as the reference asserts, ms_deform_attn_cuda.cu:28-32):
 *   value [B,S,M,D]  loc [B,Lq,M,L,P,2] (x then y)  attn [B,Lq,M,L,P]
 *   shapes [L,2] int64 (H,W)  lsi [L] int64  out / grad_out [B,Lq,M*D]
 *
 * The file is instantiated twice (double and float) through REAL/SUFFIX.
 * Accumulation order is fixed (b,q,m,l,k ascending), so results are
 * deterministic; the reference's atomicAdd order is not.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int msda_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL double
#define SUFFIX f64
#define FLOOR floor
#include "msda_oracle_body.inc"
#undef REAL
#undef SUFFIX
#undef FLOOR

#define REAL float
#define SUFFIX f32
#define FLOOR floorf
#include "msda_oracle_body.inc"
#undef REAL
#undef SUFFIX
#undef FLOOR
