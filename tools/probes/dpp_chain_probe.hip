// Micro-probe (tuning aid, not part of the library): what the serial chains of pn_fe_pitch_kernel pay per step when one
// operand of the multiply / add comes through a DPP modifier (row_shl, row_newbcast), for 1 and 2 waves per SIMD.
// Every loop body is ONE asm statement on fixed registers (16 steps), so the compiler adds nothing between the
// instructions.  Prints shader-clock ticks (s_memtime) per step per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v60", "v61", "v62", "vcc"

template <int PAT>
__global__ void probe(float *out, long long *cyc, int iters) {
  __shared__ float lds[24 * 1024];      // 96 KB: one block per CU
  for (int i = threadIdx.x; i < 24 * 1024; i += blockDim.x) lds[i] = 0.001f * (i & 255);
  __syncthreads();
  asm volatile("v_mov_b32 v60, 1.0\n v_mov_b32 v61, 0.5\n v_mov_b32 v62, 0.25\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n" ::: CLOB);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (PAT == 0) asm volatile(R16("v_mul_f32 v22, v60, v61\n v_add_f32 v20, v20, v22\n") ::: CLOB);
    if (PAT == 1) asm volatile(R16("v_mul_f32_dpp v22, v60, v61 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 v20, v20, v22\n") ::: CLOB);
    if (PAT == 2) asm volatile(R16("v_mul_f32_dpp v22, v60, v61 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 v20, v20, v22\n") ::: CLOB);
    if (PAT == 3) asm volatile(R16("v_add_f32_dpp v20, v60, v20 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n") ::: CLOB);
    if (PAT == 4) asm volatile(R16("v_mov_b32_dpp v22, v60 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 v20, v20, v22\n") ::: CLOB);
    if (PAT == 5) asm volatile(R16("v_add_f32 v20, v60, v20\n") ::: CLOB);
    if (PAT == 6) asm volatile(R16("v_mul_f32_dpp v22, v60, v61 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                   "v_mul_f32_dpp v23, v60, v62 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                   "v_add_f32 v20, v20, v22\n v_add_f32 v21, v21, v23\n"
                                   "v_cndmask_b32 v24, v61, v62, vcc\n v_cndmask_b32 v25, v62, v61, vcc\n") ::: CLOB);
    if (PAT == 7) asm volatile(R16("v_mul_f32 v22, v60, v61\n v_mul_f32 v23, v60, v62\n"
                                   "v_add_f32 v20, v20, v22\n v_add_f32 v21, v21, v23\n"
                                   "v_cndmask_b32 v24, v61, v62, vcc\n v_cndmask_b32 v25, v62, v61, vcc\n") ::: CLOB);
    if (PAT == 8) asm volatile(R16("v_add_f32_dpp v20, v60, v20 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                                   "v_subrev_f32_dpp v20, v61, v20 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                   "v_cndmask_b32 v24, v24, v20, vcc\n s_nop 0\n") ::: CLOB);
    if (PAT == 9) asm volatile(R16("v_pk_mul_f32 v[22:23], v[60:61], v[60:61]\n v_pk_add_f32 v[20:21], v[20:21], v[22:23]\n") ::: CLOB);
    if (PAT == 10) asm volatile(R16("v_mul_f32_dpp v22, v60, v61 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32 v20, v20, v22\n") ::: CLOB);
  }
  const long long t1 = __builtin_readcyclecounter();
  float r; asm volatile("v_add_f32 %0, v20, v21" : "=v"(r));
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (r == 123.456f) out[0] = r + lds[threadIdx.x];
}

template <int PAT> void run(const char *name, int steps_inst) {
  float *out; long long *cyc; hipMalloc(&out, 4); hipMalloc(&cyc, 256 * 8);
  for (int W : {1, 2}) {
    const int iters = 2000;
    probe<PAT><<<256, 256 * W>>>(out, cyc, 10); hipDeviceSynchronize();
    probe<PAT><<<256, 256 * W>>>(out, cyc, iters); hipDeviceSynchronize();
    std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-62s %d w/SIMD  %7.2f ticks/step/wave\n", name, W, (double)h[128] / iters / 16);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<5>("5: dependent v_add chain", 1);
  run<0>("0: mul + dependent add", 2);
  run<1>("1: mul_dpp row_shl + dependent add", 2);
  run<2>("2: mul_dpp row_newbcast + dependent add", 2);
  run<10>("10: mul_dpp quad_perm + dependent add", 2);
  run<3>("3: dependent add_dpp row_newbcast + s_nop 1", 2);
  run<4>("4: mov_dpp row_newbcast + dependent add", 2);
  run<7>("7: 2 mul + 2 add + 2 cndmask", 6);
  run<6>("6: 2 mul_dpp newbcast + 2 add + 2 cndmask", 6);
  run<8>("8: add_dpp, nop1, subrev_dpp, cndmask, nop0 (yy_lookup step)", 5);
  run<9>("9: pk_mul + dependent pk_add (2 chains)", 2);
  return 0;
}
