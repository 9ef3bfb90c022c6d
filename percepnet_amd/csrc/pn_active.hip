// Per-call active set (round-4 verdict item 5): pn_process_*_active advances only the streams it is given.
//
// In the reference a stream's state moves only when ITS rnnoise_process_frame is called (denoise.cpp:508-547,
// rnnoise.h:60); a batched context moves all B streams in lock-step and indexes every ring by its own global counter
// (history slot t % 12, look-ahead rings t % 6, conv FIFOs tn % 5 / tn % 3, GRU halves tn & 1).  A stream whose packet is
// late must keep ALL of its state and produce no output for that tick.
//
// How, without touching a hot kernel: every per-tick write of the frame kernels lands either
//   (a) in the ring slot that holds the OLDEST, already dead entry (history frame t-12, look-ahead spectrum t-6, fc
//       output t-5, conv1 output t-3), or in the ping-pong half the next step does not read,
//   (b) in scratch that the next frame recomputes before reading (features, silence, Ps, c2out, g|r), or
//   (c) in place: the synthesis overlap memory, last_period, last_gain — and the caller's output rows.
// So the frame runs for all rows as always (a skipped row computes on whatever its input row holds; nothing of it
// survives), bracketed by two small launches over the INACTIVE rows only:
//   before  save (c) — 4.1 KB per skipped stream;
//   after   restore (c), then shift every ring of the row by one slot, newest entry first, so that at the context's
//           next counter value the row's entries sit where the kernels will look for them (a stream that skips a tick
//           falls one slot behind the global phase; the shift re-aligns it — ≈ 52 KB moved per skipped stream-tick).
// An all-active call launches neither; the cost is paid per skipped stream.
#include "pn_common.h"
#include "pn_launch.h"

// rows ids[i] of the in-place state + the caller's output rows -> save area row i
__global__ __launch_bounds__(128) void pn_inactive_save_kernel(PnActiveArgs a) {
  const int i = blockIdx.x, s = a.ids[i], tid = threadIdx.x;
  const float4 *sy = reinterpret_cast<const float4 *>(a.synth + (size_t)s * PN_FRAME);
  float4 *dsy = reinterpret_cast<float4 *>(a.save_synth + (size_t)i * PN_FRAME);
  for (int c = tid; c < PN_FRAME / 4; c += 128) dsy[c] = sy[c];
  // the output row: 480 int16 (240 words) or 480 float (480 words)
  const uint32_t *o = reinterpret_cast<const uint32_t *>(a.out) + (size_t)s * a.out_row_words;
  uint32_t *so = a.save_out + (size_t)i * PN_FRAME;
  for (int c = tid; c < a.out_row_words; c += 128) so[c] = o[c];
  if (a.d_gr) for (int c = tid; c < 68; c += 128) a.save_gr[(size_t)i * 68 + c] = a.d_gr[(size_t)s * 68 + c];
  if (tid == 0) { a.save_period[i] = a.last_period[s]; a.save_gain[i] = a.last_gain[s]; }
}

// move k = 1..live of one ring row: src = (src0 - (k - 1)) mod slots, dst = src + 1 mod slots; column-private (a thread
// walks its own float4 column through the slots, newest first: every source is read before it is overwritten)
__device__ __forceinline__ void pn_shift_ring(float4 *row0, long long slot_stride4, int cols4, int slots, int src0, int live, int tid, int nthreads) {
  for (int c = tid; c < cols4; c += nthreads) {
    int src = src0;
    for (int k = 0; k < live; k++) {
      const int dst = src + 1 == slots ? 0 : src + 1;
      row0[(size_t)dst * slot_stride4 + c] = row0[(size_t)src * slot_stride4 + c];
      src = src == 0 ? slots - 1 : src - 1;
    }
  }
}
// the same for a row of a fragment-order operand shadow (pn_nn_x3.hip): [M tile of 128][column tile of 32][plane][k-group of
// 4][row 0..127][8 halfs]; "column" = one uint4 (8 halfs) of the row, cols = column tiles x planes x 4
__device__ __forceinline__ void pn_shift_shadow(uint4 *S, long long slot_stride_u4, int n_ct, int np, int r, int slots, int src0, int live, int tid, int nthreads) {
  if (!S) return;
  uint4 *tile = S + (size_t)(r >> 7) * n_ct * np * 512 + (r & 127);
  for (int c = tid; c < n_ct * np * 4; c += nthreads) {
    const size_t off = (size_t)(c >> 2) * 512 + (c & 3) * 128;
    int src = src0;
    for (int k = 0; k < live; k++) {
      const int dst = src + 1 == slots ? 0 : src + 1;
      tile[(size_t)dst * slot_stride_u4 + off] = tile[(size_t)src * slot_stride_u4 + off];
      src = src == 0 ? slots - 1 : src - 1;
    }
  }
}

__global__ __launch_bounds__(256) void pn_inactive_fixup_kernel(PnActiveArgs a) {
  const int i = blockIdx.x, s = a.ids[i], tid = threadIdx.x;
  // (c) in-place state and the caller's rows back
  {
    const float4 *ssy = reinterpret_cast<const float4 *>(a.save_synth + (size_t)i * PN_FRAME);
    float4 *sy = reinterpret_cast<float4 *>(a.synth + (size_t)s * PN_FRAME);
    for (int c = tid; c < PN_FRAME / 4; c += 256) sy[c] = ssy[c];
    uint32_t *o = reinterpret_cast<uint32_t *>(a.out) + (size_t)s * a.out_row_words;
    const uint32_t *so = a.save_out + (size_t)i * PN_FRAME;
    for (int c = tid; c < a.out_row_words; c += 256) o[c] = so[c];
    if (a.d_gr) for (int c = tid; c < 68; c += 256) a.d_gr[(size_t)s * 68 + c] = a.save_gr[(size_t)i * 68 + c];
    if (tid == 0) { a.last_period[s] = a.save_period[i]; a.last_gain[s] = a.save_gain[i]; }
  }
  if (a.restore_only) return;
  // (a) rings: t / tn are the counters of the tick that has just run
  const int t12 = (int)(a.t % 12), t6 = (int)(a.t % 6), n5 = (int)(a.tn % 5), n3 = (int)(a.tn % 3), n2 = (int)(a.tn & 1);
  {   // history: frames t-1 .. t-11 (slots t-1 .. t-11 mod 12) one slot up; slots sit back to back inside the row
    float4 *h = reinterpret_cast<float4 *>(a.hist + (size_t)s * PN_HIST_STRIDE);
    pn_shift_ring(h, PN_FRAME / 4, PN_FRAME / 4, 12, (t12 + 11) % 12, 11, tid, 256);
    // the mirror of the ring's first 8 samples (unaligned comb-tap loads): columns 0 and 1 of slot 0, moved by threads 0 and 1
    if (tid < 2) h[PN_HIST / 4 + tid] = h[tid];
  }
  pn_shift_ring(reinterpret_cast<float4 *>(a.yring + (size_t)s * PN_SPEC_BINS), (long long)a.B * PN_SPEC_BINS / 2, PN_SPEC_BINS / 2, 6, (t6 + 5) % 6, 5, tid, 256);
  pn_shift_ring(reinterpret_cast<float4 *>(a.eyring + (size_t)s * 36), (long long)a.B * 9, 9, 6, (t6 + 5) % 6, 5, tid, 256);
  pn_shift_ring(reinterpret_cast<float4 *>(a.c1ring + (size_t)s * 128), a.Bp * 32, 32, 5, (n5 + 4) % 5, 4, tid, 256);
  pn_shift_ring(reinterpret_cast<float4 *>(a.c2ring + (size_t)s * 512), a.Bp * 128, 128, 3, (n3 + 2) % 3, 2, tid, 256);
  // ping-pong pairs: the live state is in half tn & 1 (the half this tick read); the next tick reads the other one
  for (int g = 0; g < 4; g++) pn_shift_ring(reinterpret_cast<float4 *>(a.gru[g] + (size_t)s * 512), a.Bp * 128, 128, 2, n2, 1, tid, 256);
  pn_shift_ring(reinterpret_cast<float4 *>(a.rb + (size_t)s * 128), a.Bp * 32, 32, 2, n2, 1, tid, 256);
  // operand shadows of the fp16-operand / split-precision modes
  if (a.np) {
    pn_shift_shadow(a.c1ringH, a.np * a.Bp * 128 / 8, 4, a.np, s, 5, (n5 + 4) % 5, 4, tid, 256);
    pn_shift_shadow(a.c2ringH, a.np * a.Bp * 512 / 8, 16, a.np, s, 3, (n3 + 2) % 3, 2, tid, 256);
    for (int g = 0; g < 4; g++) pn_shift_shadow(a.gruH[g], a.np * a.Bp * 512 / 8, 16, a.np, s, 2, n2, 1, tid, 256);
    pn_shift_shadow(a.rbH, a.np * a.Bp * 128 / 8, 4, a.np, s, 2, n2, 1, tid, 256);
  }
}

void pn_launch_inactive_save(hipStream_t st, const PnActiveArgs &a, int n) {
  if (n > 0) hipLaunchKernelGGL(pn_inactive_save_kernel, dim3(n), dim3(128), 0, st, a);
}
void pn_launch_inactive_fixup(hipStream_t st, const PnActiveArgs &a, int n) {
  if (n > 0) hipLaunchKernelGGL(pn_inactive_fixup_kernel, dim3(n), dim3(256), 0, st, a);
}

// ---- a kernel that only passes time (pipe_init's queue probe, pn_context.cpp) --------------------------------------
// One wave that sleeps until `ticks` of the 100 MHz wall clock have gone by.
__global__ void pn_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
int pn_launch_spin(hipStream_t st, long long ticks) {
  hipLaunchKernelGGL(pn_spin_kernel, dim3(1), dim3(64), 0, st, ticks);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
