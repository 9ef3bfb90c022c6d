// Micro-probe (tuning aid, not part of the library), round 4: two waves per SIMD, one issuing v_mfma_f32_32x32x16_f16 back
// to back on independent accumulators (the K phase of pn_gru_x3p_kernel), the other VALU work (its gating epilogue).  Do the
// two streams overlap — and does it depend on where the accumulators live (ArchVGPRs or AccVGPRs), on the kind of VALU
// instruction (v_pk_*_f32, plain f32, integer / conversion), and on the partner's LDS reads?
// Block = 8 waves on one CU (LDS-padded to one block per CU): waves 0-3 MFMA, waves 4-7 (same SIMDs) the companion stream.
// Prints cycles per MFMA seen by the MFMA waves and cycles per instruction seen by the companions, alone and together.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float fvec4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define SB() __builtin_amdgcn_sched_barrier(0)

#define MFMA_V(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb))
#define MFMA_A(acc) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(fa), "v"(fb))

__global__ __launch_bounds__(512) void probe(float *__restrict__ out, long long *cyc, int iters, int mfma_mode, int comp_mode) {
  __shared__ float pad[30000];            // 120 KB: one block per CU
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  pad[tid] = tid * 0.001f;
  __syncthreads();
  if (wave < 4) {
    floatx16 a0, a1, a2, a3, a4, a5;
    for (int i = 0; i < 16; i++) { a0[i] = lane; a1[i] = 1; a2[i] = 2; a3[i] = 3; a4[i] = 4; a5[i] = 5; }
    fvec4 fa = {1.f, 2.f, 3.f, 4.f}, fb = {lane * 1.f, 0.f, 1.f, 2.f};
    const long long t0 = __builtin_readcyclecounter();
    if (mfma_mode == 1) {
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) { MFMA_V(a0); MFMA_V(a1); MFMA_V(a2); MFMA_V(a3); MFMA_V(a4); MFMA_V(a5); }
      }
    } else if (mfma_mode == 2) {
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) { MFMA_A(a0); MFMA_A(a1); MFMA_A(a2); MFMA_A(a3); MFMA_A(a4); MFMA_A(a5); }
      }
    } else if (mfma_mode == 3) {          // the K loop's shape: 6 MFMAs, then a ds_read_b128 whose result the next group waits for
      fvec4 r = fb;
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
          fb = r;
          r = *reinterpret_cast<const fvec4 *>(&pad[(lane * 4 + q * 256) & 16383]); SB();
          MFMA_V(a0); MFMA_V(a1); MFMA_V(a2); MFMA_V(a3); MFMA_V(a4); MFMA_V(a5); SB();
        }
      }
      fa = r;
    } else if (mfma_mode == 4) {          // barrier-coupled: 12 MFMAs, then s_barrier (4 per iteration = 48 MFMAs)
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          MFMA_V(a0); MFMA_V(a1); MFMA_V(a2); MFMA_V(a3); MFMA_V(a4); MFMA_V(a5);
          MFMA_V(a0); MFMA_V(a1); MFMA_V(a2); MFMA_V(a3); MFMA_V(a4); MFMA_V(a5); SB();
          __builtin_amdgcn_s_barrier(); SB();
        }
      }
    } else if (mfma_mode == 5) {          // barriers only (4 per iteration)
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) { __builtin_amdgcn_s_barrier(); SB(); }
      }
    } else {
      for (int it = 0; it < iters; it++) __builtin_amdgcn_s_sleep(100);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = fa.x;
    for (int i = 0; i < 16; i++) s += a0[i] + a1[i] + a2[i] + a3[i] + a4[i] + a5[i];
    out[(size_t)blockIdx.x * 512 + tid] = s;
  } else {
    v2f x0 = {lane * 1e-3f, 1.f}, x1 = {2.f, lane * 1e-3f}, x2 = {0.5f, 0.25f}, x3 = {1.5f, 0.75f};
    const v2f k = {1.0000001f, 0.9999999f};
    float y0 = lane, y1 = 1.f, y2 = 2.f, y3 = 3.f;
    int i0 = lane, i1 = 3, i2 = 5, i3 = 7;
    float acc = 0;
    float d[16];
    for (int c = 0; c < 16; c++) d[c] = lane * 0.01f + c;
    const long long t0 = __builtin_readcyclecounter();
    if (comp_mode == 11 || comp_mode == 12) {
      // tight loops (no mode dispatch inside): 480 instructions per 10 iterations' worth of the others, so that the loop
      // overhead does not hide the issue rate.  11: 16 independent v_mul_f32 chains; 12: the epilogue's mix (and / mul / add /
      // floor / cvt / med3 / lshl_add) on 16 independent chains
      for (int it = 0; it < iters / 10; it++) {
        if (comp_mode == 11) {
#pragma unroll
          for (int q = 0; q < 30; q++) {
#pragma unroll
            for (int c = 0; c < 16; c++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d[c]) : "v"(k.x));
          }
        } else {
#pragma unroll
          for (int q = 0; q < 5; q++) {
#pragma unroll
            for (int c = 0; c < 16; c++) {
              int t;
              asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(d[c]));
              asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d[c]) : "v"(k.x));
              asm volatile("v_add_f32 %0, %0, %1" : "+v"(d[c]) : "v"(k.y));
              asm volatile("v_floor_f32 %0, %1" : "=v"(y0) : "v"(d[c]));
              asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(t) : "v"(y0));
              asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(t) : "v"(200));
              i0 += t;
            }
          }
        }
      }
    } else
    // 48 instructions per iteration in every mode
    for (int it = 0; it < iters; it++) {
      if (comp_mode == 1) {               // packed f32: four independent chains
#pragma unroll
        for (int q = 0; q < 12; q++) {
          asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x0) : "v"(k)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x1) : "v"(k));
          asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x2) : "v"(k)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x3) : "v"(k));
        }
      } else if (comp_mode == 2) {        // plain f32
#pragma unroll
        for (int q = 0; q < 12; q++) {
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y0) : "v"(k.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(y1) : "v"(k.y));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y2) : "v"(k.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(y3) : "v"(k.y));
        }
      } else if (comp_mode == 3) {        // integer / bit operations
#pragma unroll
        for (int q = 0; q < 12; q++) {
          asm volatile("v_and_b32 %0, %0, %1" : "+v"(i0) : "v"(0x7fffffff)); asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(i1) : "v"(200));
          asm volatile("v_sub_u32 %0, 0, %0" : "+v"(i2)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(i3) : "v"(i0));
        }
      } else if (comp_mode == 4) {        // conversions + floor (the table-index path)
#pragma unroll
        for (int q = 0; q < 12; q++) {
          asm volatile("v_floor_f32 %0, %0" : "+v"(y0)); asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(i1) : "v"(y1));
          asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(y2) : "v"(i2)); asm volatile("v_floor_f32 %0, %0" : "+v"(y3));
        }
      } else if (comp_mode == 5) {        // LDS reads: 12 ds_read_b32 + 36 plain f32
#pragma unroll
        for (int q = 0; q < 12; q++) {
          const float v = pad[(i0 + q * 64) & 16383]; acc += v; SB();
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y0) : "v"(k.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(y1) : "v"(k.y));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y2) : "v"(k.x));
        }
      } else if (comp_mode == 6) {        // dense plain f32: 16 independent chains (issue-bound, not latency-bound)
#pragma unroll
        for (int q = 0; q < 3; q++) {
#pragma unroll
          for (int c = 0; c < 16; c++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d[c]) : "v"(k.x));
        }
      } else if (comp_mode == 7) {        // dense mix: 16 independent chains of mul / add / and / floor
#pragma unroll
        for (int q = 0; q < 3; q++) {
#pragma unroll
          for (int c = 0; c < 16; c += 4) {
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d[c]) : "v"(k.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(d[c + 1]) : "v"(k.y));
            asm volatile("v_and_b32 %0, %0, %1" : "+v"(d[c + 2]) : "v"(0x7fffffff)); asm volatile("v_floor_f32 %0, %0" : "+v"(d[c + 3]));
          }
        }
      } else if (comp_mode == 8 || comp_mode == 9 || comp_mode == 10) {   // barrier-coupled: (8) 64 / (10) 32 dense plain VALU, then s_barrier, 4x; (9) barriers only
#pragma unroll
        for (int q = 0; q < 4; q++) {
          if (comp_mode != 9) {
#pragma unroll
            for (int r = 0; r < (comp_mode == 8 ? 4 : 2); r++) {
#pragma unroll
              for (int c = 0; c < 16; c++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d[c]) : "v"(k.x));
            }
          }
          SB(); __builtin_amdgcn_s_barrier(); SB();
        }
      } else {
        __builtin_amdgcn_s_sleep(100);
      }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    for (int c = 0; c < 16; c++) acc += d[c];
    out[(size_t)blockIdx.x * 512 + tid] = x0.x + x1.x + x2.x + x3.x + y0 + y1 + y2 + y3 + i0 + i1 + i2 + i3 + acc;
  }
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000, nblk = 256;
  float *out; long long *cyc;
  hipMalloc(&out, (size_t)nblk * 512 * 4); hipMalloc(&cyc, nblk * 8 * 8);
  long long *h = (long long *)malloc(nblk * 8 * 8);
  const char *mn[6] = {"no MFMA", "MFMA acc in ArchVGPRs", "MFMA acc in AccVGPRs", "MFMA + ds_read_b128 / 6", "12 MFMA + s_barrier", "s_barrier only"};
  const char *cn[13] = {"idle", "v_pk_mul/add_f32", "v_mul/add_f32", "and/med3/sub/lshl_add", "floor/cvt", "ds_read_b32 + 3 f32",
                       "dense v_mul_f32 (16 chains)", "dense mul/add/and/floor",
                       "64 v_mul + s_barrier", "s_barrier only", "32 v_mul + s_barrier",
                       "TIGHT dense v_mul_f32 (16 chains)", "TIGHT and/mul/add/floor/cvt/med3 (16 chains)"};
  for (int mm = 0; mm < 6; mm++)
    for (int cm = 0; cm < 13; cm++) {
      if (mm == 0 && cm == 0) continue;
      if ((mm >= 4) != (cm >= 8 && cm <= 10)) continue;           // barrier-coupled modes only with each other
      for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(probe, dim3(nblk), dim3(512), 0, 0, out, cyc, iters, mm, cm);
      hipDeviceSynchronize();
      hipMemcpy(h, cyc, nblk * 8 * 8, hipMemcpyDeviceToHost);
      double sm = 0, sc = 0;
      for (int b = 0; b < nblk; b++) for (int w = 0; w < 8; w++) (w < 4 ? sm : sc) += (double)h[b * 8 + w];
      printf("%-24s | companion %-24s : %6.1f cycles per MFMA, %5.2f cycles per companion instruction\n", mn[mm], cn[cm],
             mm ? sm / (nblk * 4) / ((double)iters * 48) : 0.0, cm ? sc / (nblk * 4) / ((double)iters * 48) : 0.0);
    }
  return 0;
}
