// kbench.hip - developer micro-benchmark: times the framing/validation kernels on a synthetic
// HBM-resident batch, with ablation variants, using hipEvents.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o kbench kbench.hip && ./kbench [reads] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#include "../../fastq_utils_amd/csrc/fqg_device.h"
#include "../../fastq_utils_amd/csrc/fqg_kernels.hip"
#include "../../fastq_utils_amd/csrc/fqg_stream_kernels.hip"
#ifdef KBENCH_EXTRA
#include KBENCH_EXTRA
#endif
using namespace fqg;
// census with each lane reading 32 contiguous bytes (two 16-byte loads at a 32-byte lane stride)
__global__ __launch_bounds__(kBlock) void k_count_nl_s32(const uint8_t* __restrict__ img, uint64_t n, uint32_t n_chunks, uint32_t* __restrict__ out) {
  const uint32_t chunk = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (chunk >= n_chunks || (uint64_t)(chunk + 1) * kChunkBytes > n) return;
  const uint8_t* p = img + (uint64_t)chunk * kChunkBytes + (threadIdx.x & 63) * 32;
  uint32_t cnt = 0;
  uint4 v[4];
  v[0] = *reinterpret_cast<const uint4*>(p); v[1] = *reinterpret_cast<const uint4*>(p + 16);
  v[2] = *reinterpret_cast<const uint4*>(p + 2048); v[3] = *reinterpret_cast<const uint4*>(p + 2048 + 16);
#pragma unroll
  for (int k = 0; k < 4; ++k) cnt += __popc(eq_bytes(v[k].x, 0x0A0A0A0Au)) + __popc(eq_bytes(v[k].y, 0x0A0A0A0Au)) + __popc(eq_bytes(v[k].z, 0x0A0A0A0Au)) + __popc(eq_bytes(v[k].w, 0x0A0A0A0Au));
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) out[chunk] = cnt;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static double time_ms(int reps, const std::function<void()>& f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  const uint64_t reads = argc > 1 ? strtoull(argv[1], 0, 10) : 20000000ull;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const uint32_t L = 150;
  const uint64_t R = 45 + 2ull * (L + 1) + 2, n = reads * R;
  uint8_t* img; CK(hipMalloc(&img, n + 64));
  hipLaunchKernelGGL(k_synth, dim3(256 * 32), dim3(kBlock), 0, 0, img, reads, L, 0ull, 12345ull, 1);
  CK(hipDeviceSynchronize());
  const uint32_t n_tiles = (uint32_t)((n + kChunkBytes - 1) / kChunkBytes), n_spans = (n_tiles + kScanSpan - 1) / kScanSpan;
  uint32_t *counts, *local, *suspect; unsigned long long *spans, *list; uint64_t* line_end; CallState* cs; AccState* acc; unsigned long long* hist;
  CK(hipMalloc(&counts, n_tiles * 4ull)); CK(hipMalloc(&local, n_tiles * 4ull)); CK(hipMalloc(&spans, n_spans * 8ull));
  CK(hipMalloc(&line_end, (reads * 4 + 2) * 8)); CK(hipMalloc(&cs, sizeof(CallState))); CK(hipMalloc(&acc, sizeof(AccState)));
  CK(hipMalloc(&hist, 8ull * FQG_MAX_READ_LENGTH)); CK(hipMalloc(&suspect, (reads / 32 + 2) * 4)); CK(hipMalloc(&list, 8ull << 20));
  { CallState init{}; init.first_key = ~0ull; init.stop_record = ~0ull; init.qmin_byte = 255; CK(hipMemcpy(cs, &init, sizeof(init), hipMemcpyHostToDevice)); } CK(hipMemset(hist, 0, 8ull * FQG_MAX_READ_LENGTH)); CK(hipMemset(suspect, 0, (reads / 32 + 2) * 4));
  AccState init{0, FQG_MAX_READ_LENGTH, 0, 255, 0}; CK(hipMemcpy(acc, &init, sizeof(init), hipMemcpyHostToDevice));
  const double gb = n / 1e9;
  auto report = [&](const char* name, double ms) { printf("%-28s %8.3f ms  %8.1f GB/s (input bytes)\n", name, ms, gb / (ms * 1e-3)); };
  report("k_count_nl", time_ms(reps, [&] { hipLaunchKernelGGL(k_count_nl, dim3((n_tiles + 3) / 4), dim3(kBlock), 0, 0, img, n, n_tiles, counts, cs); }));
  hipLaunchKernelGGL(k_scan_a, dim3(n_spans), dim3(kBlock), 0, 0, counts, n_tiles, local, spans);
  hipLaunchKernelGGL(k_scan_b, dim3(1), dim3(kBlock), 0, 0, spans, n_spans, img, n, cs);
  CK(hipDeviceSynchronize());
  const unsigned grid_t = 256 * 8;
  SuspectMap sm{suspect, reads, &cs->flags};
#define FF(ABL, label) report(label, time_ms(reps, [&] { hipLaunchKernelGGL(k_frame_fast_t<ABL>, dim3(grid_t), dim3(kBlock), 0, 0, img, n, n_tiles, local, spans, line_end, 4 * reads + 32, 4 * reads, sm, cs, (const uint32_t*)nullptr, (const uint32_t*)nullptr); }))
  { CallState init{}; init.first_key = ~0ull; init.stop_record = ~0ull; init.qmin_byte = 255; init.boot_qmin = 255; init.n_newlines = 4 * reads; init.last_byte_is_nl = 1; CK(hipMemcpy(cs, &init, sizeof(init), hipMemcpyHostToDevice)); }
  FF(0u, "k_frame_fast (full)");
  FF(1u, "  - no qual range");
  FF(2u, "  - no base check");
  FF(4u, "  - no line-start check");
  FF(8u, "  - no line_end stores");
  FF(7u, "  line index only (exact path)");
  {
    // streaming (single-pass) path
    uint32_t *cinfo, *redo; uint16_t* stage; unsigned long long* queue;
    CK(hipMalloc(&cinfo, n_tiles * 4ull)); CK(hipMalloc(&redo, n_tiles * 4ull)); CK(hipMalloc(&stage, (size_t)n_tiles * kStageCap * 2)); CK(hipMalloc(&queue, 8ull << 20));
    StreamOut so{counts, cinfo, stage, queue, 1ull << 20};
    report("k_stream_boot", time_ms(reps, [&] { hipLaunchKernelGGL(k_stream_boot, dim3(1), dim3(kBootBlock), 0, 0, img, 65536u, cs); }));
#define P1(ABL, label) report(label, time_ms(reps, [&] { hipLaunchKernelGGL(k_stream_pass1<ABL>, dim3((n_tiles + 3) / 4), dim3(kBlock), 0, 0, img, n, n_tiles, so, cs); }))
    P1(1u, "  pass1 - no checks");
    P1(2u, "  pass1 - no staging");
    P1(3u, "  pass1 - masks+scan only");
    report("  census, 32B-stride loads", time_ms(reps, [&] { hipLaunchKernelGGL(k_count_nl_s32, dim3((n_tiles + 3) / 4), dim3(kBlock), 0, 0, img, n, n_tiles, cinfo); }));
    P1(4u, "  pass1 - no next-byte fetch");
    P1(8u, "  pass1 - no base check");
    P1(16u, "  pass1 - no quality test");
    P1(24u, "  pass1 - types only");
    P1(0u, "k_stream_pass1");
    hipLaunchKernelGGL(k_scan_a, dim3(n_spans), dim3(kBlock), 0, 0, counts, n_tiles, local, spans);
    hipLaunchKernelGGL(k_scan_b, dim3(1), dim3(kBlock), 0, 0, spans, n_spans, img, n, cs);
    report("k_stream_pass2", time_ms(reps, [&] { hipLaunchKernelGGL(k_stream_pass2, dim3((n_tiles + 31) / 32), dim3(kBlock), 0, 0, img, n, n_tiles, counts, cinfo, stage, local, spans, line_end, 4 * reads + 32, 4 * reads, sm, redo, cs); }));
    {
      // Does the second framing pass overlap with pass 1 when the two run on streams of their own?  (pass 1 is bound by
      // vector-ALU cycles, k_stream_lines_fast by memory latency - but pass 1 fills every wave slot and most of the LDS.)
      // The arrays the line kernel reads are those of the run above; the pass 1 beside it writes to a second set.
      uint32_t *counts2, *cinfo2; uint16_t* stage2;
      CK(hipMalloc(&counts2, n_tiles * 4ull)); CK(hipMalloc(&cinfo2, n_tiles * 4ull)); CK(hipMalloc(&stage2, (size_t)n_tiles * kStageCap * 2));
      StreamOut so2{counts2, cinfo2, stage2, queue, 1ull << 20};
      LinesArgs A{};
      A.img = img; A.n = n; A.cr = ChunkRanks{counts, local, spans, n_tiles}; A.stage = stage; A.line_end = line_end;
      A.line_cap = 4 * reads + 32; A.n_newlines = 4 * reads; A.n_lines = 4 * reads; A.limit = 4 * reads; A.suspect_bits = suspect;
      A.suspect_cap = reads; A.flags = &cs->flags; A.space = 0; A.weight = 1; A.acc = acc; A.hist = hist; A.ablate = 0;
      const uint64_t groups = (A.n_lines + 4 * kWave - 1) / (4 * kWave), n_steps = (groups + kLinesPer - 1) / kLinesPer;
      uint8_t* todo; CK(hipMalloc(&todo, n_steps + 4)); CK(hipMemset(todo, 0, n_steps + 4));
      int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_stream_lines_fast), kBlock, 0));
      const unsigned grid_f = (unsigned)std::min<uint64_t>((n_steps + 3) / 4, 256ull * nb);
      report("k_stream_lines_fast", time_ms(reps, [&] { hipLaunchKernelGGL(k_stream_lines_fast, dim3(grid_f), dim3(kBlock), 0, 0, A, todo); }));
      int lo_p = 0, hi_p = 0; CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
      hipStream_t sa, sb, sb0; CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo_p)); CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi_p));
      CK(hipStreamCreateWithPriority(&sb0, hipStreamNonBlocking, lo_p));
      hipEvent_t e0, e1, eb; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&eb));
      auto both = [&](hipStream_t second, int order, const char* label) {
        double tot = 0;
        for (int i = 0; i <= reps; ++i) {
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0, sa));
          CK(hipStreamWaitEvent(second, e0, 0));
          if (order == 0) hipLaunchKernelGGL(k_stream_pass1<0u>, dim3((n_tiles + 3) / 4), dim3(kBlock), 0, sa, img, n, n_tiles, so2, cs);
          hipLaunchKernelGGL(k_stream_lines_fast, dim3(grid_f), dim3(kBlock), 0, second, A, todo);
          if (order == 1) hipLaunchKernelGGL(k_stream_pass1<0u>, dim3((n_tiles + 3) / 4), dim3(kBlock), 0, sa, img, n, n_tiles, so2, cs);
          CK(hipEventRecord(eb, second));
          CK(hipStreamWaitEvent(sa, eb, 0));
          CK(hipEventRecord(e1, sa));
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (i) tot += ms;
        }
        report(label, tot / reps);
      };
      printf("stream priorities: least %d, greatest %d\n", lo_p, hi_p);
      both(sb0, 0, "pass1 || lines_fast, equal prio");
      both(sb, 0, "pass1 || lines_fast, lines high");
      both(sb, 1, "lines_fast first, then pass1");
      // pass 1 in four launches, the line kernel beside the second
      {
        double tot = 0;
        const uint32_t per = ((n_tiles / 4 + 3) / 4) * 4;
        for (int i = 0; i <= reps; ++i) {
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0, sa));
          for (int sgm = 0; sgm < 4; ++sgm) {
            const uint32_t c0 = sgm * per, c1 = sgm == 3 ? n_tiles : (sgm + 1) * per;
            // (a launch over the first c1 chunks whose first c0 / 4 workgroups return at once would skew the timing: the kernel takes no offset)
            hipLaunchKernelGGL(k_stream_pass1<0u>, dim3((c1 - c0 + 3) / 4), dim3(kBlock), 0, sa, img + (uint64_t)c0 * kChunkBytes, n - (uint64_t)c0 * kChunkBytes, c1 - c0, so2, cs);
            if (sgm == 0) { CK(hipEventRecord(eb, sa)); CK(hipStreamWaitEvent(sb, eb, 0)); hipLaunchKernelGGL(k_stream_lines_fast, dim3(grid_f), dim3(kBlock), 0, sb, A, todo); }
          }
          CK(hipEventRecord(eb, sb));
          CK(hipStreamWaitEvent(sa, eb, 0));
          CK(hipEventRecord(e1, sa));
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (i) tot += ms;
        }
        report("pass1 x4, lines beside 2..4", tot / reps);
      }
    }
    CallState h; CK(hipMemcpy(&h, cs, sizeof(h), hipMemcpyDeviceToHost));
    printf("stream: flags=%u queue=%llu redo=%u (of %u x %d launches) boot q=[%u,%u]\n", h.flags, h.queue_count, h.redo_count, n_tiles, reps + 1, h.boot_qmin, h.boot_qmax);
  }
#ifdef KBENCH_EXTRA
  kbench_extra(img, n, reads, n_tiles, counts, local, spans, line_end, suspect, acc, cs, hist, list, reps, report);
#endif
  FrameView fv{img, n, line_end, reads * 4, reads};
  report("k_records_fast", time_ms(reps, [&] { hipLaunchKernelGGL(k_records_fast, dim3(256 * 8), dim3(kBlock), 0, 0, fv, 0, 1u, sm, list, 1ull << 20, &cs->list_count, acc, hist, cs); }));
  CallState h; CK(hipMemcpy(&h, cs, sizeof(h), hipMemcpyDeviceToHost));
  AccState ha; CK(hipMemcpy(&ha, acc, sizeof(ha), hipMemcpyDeviceToHost));
  printf("check: newlines=%llu (want %llu) suspects=%llu q=[%u,%u] acc q=[%u,%u] rl=[%llu,%llu]\n", h.n_newlines, (unsigned long long)reads * 4, h.list_count, h.qmin_byte, h.qmax_byte, ha.min_qbyte, ha.max_qbyte, ha.min_rl, ha.max_rl);
  return 0;
}
