// fqg_abi.hip - the C-ABI of libfqgpu.so (include/fqg.h): contexts, workspaces, launches.
// Single translation unit: the kernels are included so that launch sites see them directly.
#include <hip/hip_runtime.h>

#include <cstring>  // (before rocPRIM, whose texture iterator calls memset unqualified)
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fqg_device.h"
#include "fqg_kernels.hip"
#include "fqg_stream_kernels.hip"
#include "fqg_index_kernels.hip"
#include "fqg_names_build_kernels.hip"
#include "fqg_barcode_kernels.hip"
#include "fqg_census_kernels.hip"
#include "fqg_filter_kernels.hip"
#include "fqg_umi_kernels.hip"
#include "fqg_umi_rl_kernels.hip"
#include "fqg_umi_cell_kernels.hip"
#include "fqg_bamtags_kernels.hip"

using namespace fqg;

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct ProfEvent {
  int slot;
  hipEvent_t a, b;
};

struct ProfSlot {
  std::string name;
  uint64_t launches = 0;
  double ms = 0;
};

// Ablation switches (FQGPU_LINES_ABL, FQGPU_NAMES_ABL, FQGPU_BC_ABL, FQGPU_UMI_INSERT_ABL: kernels that skip a part
// of their work and give knowingly WRONG results, for tools/*_abl.sh) exist only in a library built with
// -DFQG_MEASURE (`make MEASURE=1`).  The shipped library does not read these variables at all: a drop-in program
// that inherits one of them must not change what it computes.
// (the "every record met once" self-checks are never bypassed in the shipped library)
inline bool measured_wrong(int ablate) {
#ifdef FQG_MEASURE
  return ablate != 0;
#else
  (void)ablate;
  return false;
#endif
}
// a tuning knob that changes HOW a result is computed, never the result (A/B runs)
inline int env_int_early(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
inline int measure_int(const char* name) {
#ifdef FQG_MEASURE
  const char* e = getenv(name);
  const int v = e ? atoi(e) : 0;
  if (v) fprintf(stderr, "libfqgpu (FQG_MEASURE build): %s=%d - results of this call are NOT valid\n", name, v);
  return v;
#else
  (void)name;
  return 0;
#endif
}

}  // namespace

struct fqg_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t out_stream = nullptr;   // fqg_barcodes_output_begin: device-to-host copies beside the next piece's upload
  hipEvent_t out_ready = nullptr;     // ... recorded on `stream` where the output was produced
  bool out_pending = false;
  hipStream_t stream = nullptr;
  std::string err;
  int cu_count = 256;
  int lines_per_cu = 0;  // workgroups of k_stream_lines a CU holds (asked once)
  int lines_fast_per_cu = 0;  // ... of k_stream_lines_fast
  uint64_t stream_min = 1ull << 20;  // images at least this large take the single-pass framing path (FQGPU_STREAM_MIN)

  DevBuf image;       // staging for host images
  DevBuf tile_counts; // u32 per tile
  DevBuf tile_local;  // u32 per tile
  DevBuf span_sums;   // u64 per span
  DevBuf line_end;    // u64 per line (+1)
  DevBuf records;     // fqg_record staging for fqg_frame_records
  DevBuf suspect;     // 1 bit per record: the fast path could not vouch for it
  DevBuf list;        // u64 record indices queued for the exact validator
  DevBuf stage;       // streaming path: u16 staged newline entries, kStageCap per chunk
  DevBuf cinfo;       // streaming path: u32 info word per chunk
  DevBuf queue;       // streaming path: u64 suspect byte positions
  DevBuf redo;        // streaming path: u32 chunks whose checks are repeated with the true rank
  DevBuf lines_slow;  // streaming path: one byte per step that k_stream_lines_fast leaves to the general kernel
  // The line index on demand (LinesArgs::no_index): the streaming path's line kernels have checked the records but
  // stored only the index's tail; `args` are the arguments of the run that stores it all (index_now).
  struct LazyIndex {
    bool pending = false;
    bool fast = false;  // k_stream_lines_fast + the steps it marked, or the general kernel alone
    unsigned grid = 0, grid_f = 0;
    LinesArgs args;
  } lazy;
  // the streaming pass in parts (k_stream_pass1_lines): the line workers' statistics until the image has passed
  AccState* pipe_acc = nullptr;
  unsigned long long* pipe_hist = nullptr;
  bool pipe_dirty = false;  // a parted pass left without its merge (an error in between): cleared by the next one
  uint64_t bc_status_valid = 0;  // iterations of the last fqg_barcodes_transform whose status bytes stand (fqg_barcodes_census)
  DevBuf build_keys[2], build_cursor, build_spill;  // the name table built in LDS (fqg_names_build_kernels.hip)
  DevBuf name_recs;   // streaming path with FQG_VALIDATE_NAMES: 64-byte header records, K per chunk (NameCapture)
  DevBuf name_hcount; // ... and the headers every chunk saw
  DevBuf name_redo, name_redo_chunks;  // what the capture-fed name kernel leaves to the line-index one
  NamesView names{};            // what the name kernels need of the last capture
  uint64_t last_names_captured = 0;    // of the last index call: records whose name came straight from a capture record
  const uint8_t* names_img = nullptr;  // the image it belongs to (null: no capture) - the current frame's, or the
  uint64_t names_nbytes = 0;           // capture is not used
  DevBuf umi_names, umi_cells, umi_umis, umi_entries[2];  // results of the last fqg_umi_count
  uint64_t umi_n_umis = 0;
  uint64_t umi_n_features = 0, umi_n_cells = 0, umi_n_entries[2] = {0, 0};
  void* umi_state = nullptr;  // UmiState of a deferred fqg_umi_count (fqg_umi_abi.inc)
  DevBuf umi_arena;           // scratch memory of fqg_umi_count, kept between calls
  size_t umi_arena_wanted = 0;
  CallState* d_cs = nullptr;
  CallState* h_cs = nullptr;  // pinned
  uint64_t* h_scalar = nullptr;  // pinned, 8 x u64

  FrameView frame{};
  bool frame_valid = false;
  bool frame_img_owned = false;  // the frame's image lives in `image` (host input), not with the caller
  bool frame_borrowed = false;   // the current frame is a retained frame made current again (fqg_frame_make_current)
  uint32_t frame_flags = 0;      // kFlagNul / kFlagCr of the framed image
  DevBuf bc_status, bc_len[3], bc_off[3], bc_sum[3], bc_out[3], bc_tile_big;
  BcCall* d_bcall = nullptr;
  BcCall* h_bcall = nullptr;  // pinned
  uint64_t bc_out_bytes[3] = {0, 0, 0};
  DevBuf bt_in, bt_off, bt_tables, bt_rec, bt_size, bt_local, bt_sums, bt_call, bt_out;  // fqg_bam_add_tags
  uint64_t bt_out_bytes = 0;
  IndexCall* d_icall = nullptr;
  IndexCall* h_icall = nullptr;  // pinned

  bool profiling = false;
  std::vector<ProfSlot> slots;
  std::vector<ProfEvent> pending;
  std::vector<hipEvent_t> free_events;
};

struct fqg_acc {
  fqg_ctx* ctx = nullptr;
  AccState* d_state = nullptr;
  unsigned long long* d_hist = nullptr;  // FQG_MAX_READ_LENGTH bins
  std::vector<unsigned long long> h_hist;
};

namespace {

int fail(fqg_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  return code;
}

#define HIP_TRY(c, call)                                              \
  do {                                                                \
    hipError_t e__ = (call);                                          \
    if (e__ != hipSuccess) return fail((c), FQG_ERR_HIP, #call, e__); \
  } while (0)

int ensure(fqg_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return 0;
  if (b.p) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, FQG_ERR_HIP, "hipStreamSynchronize", e);
    (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
  }
  size_t want = bytes + bytes / 8 + 256;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    want = bytes;
    e = hipMalloc(&b.p, want);
  }
  if (e != hipSuccess) return fail(c, FQG_ERR_NOMEM, "hipMalloc", e);
  b.cap = want;
  return 0;
}

void release(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

int prof_slot(fqg_ctx* c, const char* name) {
  for (size_t i = 0; i < c->slots.size(); ++i)
    if (c->slots[i].name == name) return (int)i;
  ProfSlot s;
  s.name = name;
  c->slots.push_back(s);
  return (int)c->slots.size() - 1;
}

hipEvent_t get_event(fqg_ctx* c) {
  if (!c->free_events.empty()) {
    hipEvent_t e = c->free_events.back();
    c->free_events.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

// fold finished event pairs into the per-kernel totals (call after the stream is idle)
void prof_drain(fqg_ctx* c) {
  for (auto& p : c->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      c->slots[p.slot].launches++;
      c->slots[p.slot].ms += ms;
    }
    c->free_events.push_back(p.a);
    c->free_events.push_back(p.b);
  }
  c->pending.clear();
}

struct ProfScope {
  fqg_ctx* c;
  ProfEvent ev{};
  bool on;
  ProfScope(fqg_ctx* ctx, const char* name) : c(ctx), on(ctx->profiling) {
    if (!on) return;
    ev.slot = prof_slot(c, name);
    ev.a = get_event(c);
    ev.b = get_event(c);
    (void)hipEventRecord(ev.a, c->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(ev.b, c->stream);
    c->pending.push_back(ev);
  }
};

void umi_drop_state(fqg_ctx* c);  // fqg_umi_abi.inc

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the FUNCTION ON A DEVICE, not to a context: two contexts of one
// device (FQGPU_DEVICES=0,0,0) share it.  The largest size asked for so far is kept per (device, kernel) for the whole
// process and only ever raised - a context that needs less never lowers what another one is about to launch with - and
// the attribute is set when (and only when) the mark moves, not per call.
int raise_dynamic_lds(fqg_ctx* c, const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::vector<std::pair<std::pair<int, const void*>, size_t>> marks;
  std::lock_guard<std::mutex> lock(mu);
  size_t* mark = nullptr;
  for (auto& m : marks)
    if (m.first.first == c->device && m.first.second == kernel) mark = &m.second;
  if (!mark) {
    marks.push_back({{c->device, kernel}, 0});
    mark = &marks.back().second;
  }
  if (bytes <= *mark) return 0;
  HIP_TRY(c, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  *mark = bytes;
  return 0;
}

int grid_for_waves(fqg_ctx* c, uint64_t n_records) {
  // persistent wave-per-record kernels: enough workgroups to fill every CU 8 deep
  uint64_t want = (n_records + (kBlock / kWave) - 1) / (kBlock / kWave);
  uint64_t cap = (uint64_t)c->cu_count * 8;
  return (int)std::max<uint64_t>(1, std::min(want, cap));
}

}  // namespace

extern "C" {

int fqg_abi_version(void) { return FQG_ABI_VERSION; }

// The runtime prepares its path for copies of more than a few KiB when the first one is submitted (13 ms,
// tools/kbench/firstcopy.hip), and threads that submit their first ones at the same moment - the contexts of
// FQGPU_DEVICES, a thread each, every one with the first piece of a file - can die inside that preparation or leave a
// copy that never completes: 27 of 2 600 runs of one fuzz-campaign case ended with SIGSEGV below hipMemcpyAsync and
// one hung in hipStreamSynchronize, none of 3 900 with such a copy made beforehand by ONE thread
// (tools/repro_campaign_case.py, profiles/r05t_first_copy_race.txt).  So when a process opens a SECOND context - the
// only way two threads can be copying at once: a context is one thread's at a time - the opening thread makes that
// first copy on every device that has a context, before it hands anything to other threads.  (A process with one
// context does not pay for it: most never copy that much.)  FQGPU_NO_FIRST_COPY=1 leaves it out, for that tool.
static void first_large_copy(int device) {
  static std::mutex mu;
  static std::vector<int> opened, copied;  // devices, each once
  static unsigned n_opens = 0;
  static const bool off = getenv("FQGPU_NO_FIRST_COPY") != nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (std::find(opened.begin(), opened.end(), device) == opened.end()) opened.push_back(device);
  if (off || ++n_opens < 2) return;
  constexpr size_t kBytes = 256u << 10;
  for (int dev : opened) {
    if (std::find(copied.begin(), copied.end(), dev) != copied.end()) continue;
    copied.push_back(dev);
    void *h = nullptr, *d = nullptr;
    hipStream_t st = nullptr;
    if (hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
        hipHostMalloc(&h, kBytes, hipHostMallocPortable) == hipSuccess && hipMalloc(&d, kBytes) == hipSuccess) {
      (void)hipMemcpyAsync(d, h, kBytes, hipMemcpyHostToDevice, st);
      (void)hipMemcpyAsync(h, d, kBytes, hipMemcpyDeviceToHost, st);
      (void)hipStreamSynchronize(st);
    }
    if (h) (void)hipHostFree(h);
    if (d) (void)hipFree(d);
    if (st) (void)hipStreamDestroy(st);
  }
  (void)hipSetDevice(device);
}

int fqg_open(int device_ordinal, fqg_ctx** out) {
  if (!out) return FQG_ERR_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FQG_ERR_NO_DEVICE;
  if (device_ordinal < 0 || device_ordinal >= n) return FQG_ERR_ARG;
  if (hipSetDevice(device_ordinal) != hipSuccess) return FQG_ERR_NO_DEVICE;
  fqg_ctx* c = new fqg_ctx();
  c->device = device_ordinal;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) c->cu_count = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return FQG_ERR_HIP;
  }
  c->stream = c->own_stream;
  if (const char* e = getenv("FQGPU_STREAM_MIN")) c->stream_min = std::max<uint64_t>(256, strtoull(e, nullptr, 10));
  if (hipMalloc((void**)&c->d_bcall, sizeof(BcCall) + 64) != hipSuccess ||
      hipHostMalloc((void**)&c->h_bcall, sizeof(BcCall) + 64, hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void**)&c->d_icall, sizeof(IndexCall)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_icall, sizeof(IndexCall), hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void**)&c->d_cs, sizeof(CallState)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_cs, sizeof(CallState), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&c->h_scalar, 8 * sizeof(uint64_t), hipHostMallocDefault) != hipSuccess) {
    fqg_close(c);
    return FQG_ERR_NOMEM;
  }
  first_large_copy(device_ordinal);
  *out = c;
  return 0;
}

void fqg_close(fqg_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  prof_drain(c);
  umi_drop_state(c);
  for (auto e : c->free_events) (void)hipEventDestroy(e);
  release(c->image);
  release(c->tile_counts);
  release(c->tile_local);
  release(c->span_sums);
  release(c->line_end);
  release(c->records);
  release(c->suspect);
  release(c->list);
  release(c->stage);
  release(c->cinfo);
  release(c->queue);
  release(c->redo);
  release(c->lines_slow);
  if (c->pipe_acc) (void)hipFree(c->pipe_acc);
  if (c->pipe_hist) (void)hipFree(c->pipe_hist);
  release(c->build_keys[0]);
  release(c->build_keys[1]);
  release(c->build_cursor);
  release(c->build_spill);
  release(c->name_recs);
  release(c->name_hcount);
  release(c->name_redo);
  release(c->name_redo_chunks);
  release(c->umi_arena);
  release(c->umi_names);
  release(c->umi_cells);
  release(c->umi_umis);
  release(c->umi_entries[0]);
  release(c->umi_entries[1]);
  release(c->bc_status);
  release(c->bc_tile_big);
  for (DevBuf* b : {&c->bt_in, &c->bt_off, &c->bt_tables, &c->bt_rec, &c->bt_size, &c->bt_local, &c->bt_sums, &c->bt_call,
                    &c->bt_out})
    release(*b);
  for (int i = 0; i < 3; ++i) {
    release(c->bc_len[i]);
    release(c->bc_off[i]);
    release(c->bc_sum[i]);
    release(c->bc_out[i]);
  }
  if (c->d_bcall) (void)hipFree(c->d_bcall);
  if (c->h_bcall) (void)hipHostFree(c->h_bcall);
  if (c->d_icall) (void)hipFree(c->d_icall);
  if (c->h_icall) (void)hipHostFree(c->h_icall);
  if (c->d_cs) (void)hipFree(c->d_cs);
  if (c->h_cs) (void)hipHostFree(c->h_cs);
  if (c->h_scalar) (void)hipHostFree(c->h_scalar);
  if (c->out_stream) {
    (void)hipStreamSynchronize(c->out_stream);
    (void)hipStreamDestroy(c->out_stream);
  }
  if (c->out_ready) (void)hipEventDestroy(c->out_ready);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* fqg_last_error(const fqg_ctx* c) { return c ? c->err.c_str() : "no context"; }

int fqg_set_stream(fqg_ctx* c, void* s) {
  if (!c) return FQG_ERR_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_drain(c);
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return 0;
}

int fqg_synchronize(fqg_ctx* c) {
  if (!c) return FQG_ERR_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

void* fqg_host_alloc(fqg_ctx* c, size_t bytes) {
  void* p = nullptr;
  if (!c || hipSetDevice(c->device) != hipSuccess) return nullptr;  // (callable from the programs' reader threads)
  // (portable: the pieces one reader thread fills are handed to whichever device's context is free, host/fq_multi.h)
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) return nullptr;
  return p;
}
void fqg_host_free(fqg_ctx* c, void* p) {
  (void)c;
  if (p) (void)hipHostFree(p);
}

// Everything the context keeps between calls only so that the next call does not allocate again - the framing buffers
// of the largest image seen, capture records, the tile kernels' plan and output text - given back to the device.
// The current frame goes with them (retained frames own their buffers and stay).
int fqg_release_scratch(fqg_ctx* c) {
  if (!c) return FQG_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->out_pending) {
    const int rcw = fqg_barcodes_output_wait(c);
    if (rcw) return rcw;
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (DevBuf* b : {&c->image, &c->tile_counts, &c->tile_local, &c->span_sums, &c->line_end, &c->records, &c->suspect, &c->list,
                    &c->stage, &c->cinfo, &c->queue, &c->redo, &c->lines_slow, &c->build_keys[0], &c->build_keys[1],
                    &c->build_cursor, &c->build_spill, &c->name_recs, &c->name_hcount, &c->name_redo, &c->name_redo_chunks,
                    &c->bc_status, &c->bc_tile_big})
    release(*b);
  for (int i = 0; i < 3; ++i) {
    release(c->bc_len[i]);
    release(c->bc_off[i]);
    release(c->bc_sum[i]);
    release(c->bc_out[i]);
    c->bc_out_bytes[i] = 0;
  }
  c->frame_valid = false;
  c->frame_borrowed = false;
  c->lazy.pending = false;
  c->names_img = nullptr;
  c->bc_status_valid = 0;
  return 0;
}

// ---- accumulator --------------------------------------------------------------------------
int fqg_acc_reset(fqg_acc* a) {
  if (!a) return FQG_ERR_ARG;
  fqg_ctx* c = a->ctx;
  AccState init;
  init.num_rds = 0;
  init.min_rl = FQG_MAX_READ_LENGTH;
  init.max_rl = 0;
  init.min_qbyte = 255;
  init.max_qbyte = 0;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(a->d_state, &init, sizeof(init), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemset(a->d_hist, 0, sizeof(unsigned long long) * FQG_MAX_READ_LENGTH));
  return 0;
}

int fqg_acc_create(fqg_ctx* c, fqg_acc** out) {
  if (!c || !out) return FQG_ERR_ARG;
  fqg_acc* a = new fqg_acc();
  a->ctx = c;
  if (hipSetDevice(c->device) != hipSuccess ||
      hipMalloc((void**)&a->d_state, sizeof(AccState)) != hipSuccess ||
      hipMalloc((void**)&a->d_hist, sizeof(unsigned long long) * FQG_MAX_READ_LENGTH) != hipSuccess) {
    fqg_acc_destroy(a);
    return fail(c, FQG_ERR_NOMEM, "accumulator allocation");
  }
  int rc = fqg_acc_reset(a);
  if (rc) {
    fqg_acc_destroy(a);
    return rc;
  }
  *out = a;
  return 0;
}

void fqg_acc_destroy(fqg_acc* a) {
  if (!a) return;
  if (a->ctx) (void)hipSetDevice(a->ctx->device);
  if (a->d_state) (void)hipFree(a->d_state);
  if (a->d_hist) (void)hipFree(a->d_hist);
  delete a;
}

static int acc_fetch_state(fqg_acc* a, AccState* s) {
  fqg_ctx* c = a->ctx;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(s, a->d_state, sizeof(AccState), hipMemcpyDeviceToHost));
  return 0;
}

// the reference widens each quality char through (unsigned int): bytes >= 0x80 sign-extend
static uint64_t widen_qual(unsigned b) { return b < 128 ? b : (0xFFFFFF00ull | b); }

int fqg_acc_read(fqg_acc* a, fqg_file_stats* out) {
  if (!a || !out) return FQG_ERR_ARG;
  AccState s;
  int rc = acc_fetch_state(a, &s);
  if (rc) return rc;
  out->num_rds = s.num_rds;
  out->min_rl = s.min_rl;
  out->max_rl = s.max_rl;
  // FASTQ_FILE starts at min_qual=126, max_qual=0 (src/fastq.c:174-175)
  if (s.min_qbyte <= s.max_qbyte) {
    out->min_qual = std::min<uint64_t>(FQG_MAX_PHRED_QUAL, widen_qual(s.min_qbyte));
    out->max_qual = widen_qual(s.max_qbyte);
  } else {
    out->min_qual = FQG_MAX_PHRED_QUAL;
    out->max_qual = 0;
  }
  return 0;
}

static int acc_fetch_hist(fqg_acc* a) {
  fqg_ctx* c = a->ctx;
  a->h_hist.resize(FQG_MAX_READ_LENGTH);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(a->h_hist.data(), a->d_hist, sizeof(unsigned long long) * FQG_MAX_READ_LENGTH,
                       hipMemcpyDeviceToHost));
  return 0;
}

int fqg_acc_hist_nonzero(fqg_acc* a, uint64_t* lens, uint64_t* counts, size_t cap, size_t* n) {
  if (!a || !n) return FQG_ERR_ARG;
  int rc = acc_fetch_hist(a);
  if (rc) return rc;
  size_t k = 0;
  for (size_t i = 0; i < (size_t)FQG_MAX_READ_LENGTH; ++i)
    if (a->h_hist[i]) {
      if (k < cap && lens && counts) {
        lens[k] = i;
        counts[k] = a->h_hist[i];
      }
      ++k;
    }
  *n = k;
  return 0;
}

// median_rl(), reference src/fastq_info.c:39-55
int fqg_acc_median(fqg_acc* a, fqg_acc* b, uint64_t* median) {
  if (!a || !median) return FQG_ERR_ARG;
  AccState sa, sb;
  int rc = acc_fetch_state(a, &sa);
  if (rc) return rc;
  if (sa.num_rds == 1 && !b) {
    *median = sa.min_rl;
    return 0;
  }
  uint64_t nreads = sa.num_rds;
  if ((rc = acc_fetch_hist(a))) return rc;
  if (b) {
    if ((rc = acc_fetch_state(b, &sb))) return rc;
    if ((rc = acc_fetch_hist(b))) return rc;
    nreads += sb.num_rds;
  }
  unsigned long long ctr = 0;
  uint64_t crl = 1;
  while (crl < FQG_MAX_READ_LENGTH) {
    ctr += a->h_hist[crl];
    if (b) ctr += b->h_hist[crl];
    if (sa.num_rds > 1 && ctr > nreads / 2) break;
    ++crl;
  }
  *median = crl;
  return 0;
}

// export layout: AccState | u64 n | n x (u64 len, u64 count)
int fqg_acc_export(fqg_acc* a, void* buf, size_t cap, size_t* used) {
  if (!a || !used) return FQG_ERR_ARG;
  AccState s;
  int rc = acc_fetch_state(a, &s);
  if (rc) return rc;
  if ((rc = acc_fetch_hist(a))) return rc;
  std::vector<uint64_t> pairs;
  for (size_t i = 0; i < (size_t)FQG_MAX_READ_LENGTH; ++i)
    if (a->h_hist[i]) {
      pairs.push_back(i);
      pairs.push_back(a->h_hist[i]);
    }
  const size_t need = sizeof(AccState) + 8 + pairs.size() * 8;
  *used = need;
  if (!buf || cap < need) return buf ? FQG_ERR_ARG : 0;
  char* w = (char*)buf;
  memcpy(w, &s, sizeof(s));
  const uint64_t n = pairs.size() / 2;
  memcpy(w + sizeof(s), &n, 8);
  if (n) memcpy(w + sizeof(s) + 8, pairs.data(), pairs.size() * 8);
  return 0;
}

int fqg_acc_merge(fqg_acc* a, const void* buf, size_t used) {
  if (!a || !buf || used < sizeof(AccState) + 8) return FQG_ERR_ARG;
  fqg_ctx* c = a->ctx;
  AccState mine, other;
  int rc = acc_fetch_state(a, &mine);
  if (rc) return rc;
  const char* r = (const char*)buf;
  memcpy(&other, r, sizeof(other));
  uint64_t n;
  memcpy(&n, r + sizeof(other), 8);
  if (used < sizeof(AccState) + 8 + n * 16) return FQG_ERR_ARG;
  mine.num_rds += other.num_rds;
  mine.min_rl = std::min(mine.min_rl, other.min_rl);
  mine.max_rl = std::max(mine.max_rl, other.max_rl);
  mine.min_qbyte = std::min(mine.min_qbyte, other.min_qbyte);
  mine.max_qbyte = std::max(mine.max_qbyte, other.max_qbyte);
  HIP_TRY(c, hipMemcpy(a->d_state, &mine, sizeof(mine), hipMemcpyHostToDevice));
  if (n) {
    if ((rc = acc_fetch_hist(a))) return rc;
    const uint64_t* pr = (const uint64_t*)(r + sizeof(other) + 8);
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t len = pr[2 * i], cnt = pr[2 * i + 1];
      if (len >= (uint64_t)FQG_MAX_READ_LENGTH) return FQG_ERR_ARG;
      a->h_hist[len] += cnt;
      HIP_TRY(c, hipMemcpy(a->d_hist + len, &a->h_hist[len], 8, hipMemcpyHostToDevice));
    }
  }
  return 0;
}

// ---- framing + validation -------------------------------------------------------------------
namespace {

struct Framed {
  uint64_t n_newlines = 0;
  bool last_nl = false;
  uint32_t img_flags = 0;
  bool checks_done = false;  // the byte-class checks of the tiled path ran over the image
  bool records_done = false; // lengths, statistics and suspect bits of every record are done too (k_stream_lines)
};

void init_call_state(fqg_ctx* c) {
  CallState init;
  memset(&init, 0, sizeof(init));
  init.first_key = ~0ull;
  init.stop_record = ~0ull;
  init.trunc_record = ~0ull;
  init.qmin_byte = 255;
  init.boot_qmin = 255;
  *c->h_cs = init;
}

// Two passes over the image: newline census per chunk, prefix over the counts, then the line
// index (with or without the byte-class checks).  Sizes every buffer exactly.
int frame_two_pass(fqg_ctx* c, const uint8_t* d_img, uint64_t nbytes, uint32_t n_chunks, bool final,
                   bool want_checks, SuspectMap sm, Framed* out) {
  int rc;
  const uint32_t n_spans = (n_chunks + kScanSpan - 1) / kScanSpan;
  if ((rc = ensure(c, c->tile_counts, (size_t)n_chunks * 4))) return rc;
  if ((rc = ensure(c, c->tile_local, (size_t)n_chunks * 4))) return rc;
  if ((rc = ensure(c, c->span_sums, (size_t)n_spans * 8))) return rc;
  init_call_state(c);
  HIP_TRY(c, hipMemcpyAsync(c->d_cs, c->h_cs, sizeof(CallState), hipMemcpyHostToDevice, c->stream));
  {
    ProfScope ps(c, "k_count_nl");
    hipLaunchKernelGGL(k_count_nl, dim3((n_chunks + 3) / 4), dim3(kBlock), 0, c->stream, d_img, nbytes, n_chunks,
                       (uint32_t*)c->tile_counts.p, c->d_cs);
  }
  {
    ProfScope ps(c, "k_scan");
    hipLaunchKernelGGL(k_scan_a, dim3(n_spans), dim3(kBlock), 0, c->stream, (const uint32_t*)c->tile_counts.p,
                       n_chunks, (uint32_t*)c->tile_local.p, (unsigned long long*)c->span_sums.p);
    hipLaunchKernelGGL(k_scan_b, dim3(1), dim3(kBlock), 0, c->stream, (unsigned long long*)c->span_sums.p,
                       n_spans, d_img, nbytes, c->d_cs);
  }
  HIP_TRY(c, hipMemcpyAsync(c->h_cs, c->d_cs, sizeof(CallState), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  out->n_newlines = c->h_cs->n_newlines;
  out->last_nl = c->h_cs->last_byte_is_nl != 0;
  out->img_flags = c->h_cs->flags;
  const uint64_t n_lines_all = out->n_newlines + (out->last_nl ? 0 : 1);
  const uint64_t usable = (final || out->last_nl) ? n_lines_all : out->n_newlines;
  const uint64_t line_cap = n_lines_all + 17;
  if ((rc = ensure(c, c->line_end, (size_t)line_cap * 8))) return rc;
  const bool checks = want_checks && !(out->img_flags & (kFlagNul | kFlagCr));
  const unsigned grid = (unsigned)std::min<uint64_t>((n_chunks + 3) / 4, (uint64_t)c->cu_count * 8);
  ProfScope ps(c, checks ? "k_frame_fast" : "k_lines");
  if (checks)
    hipLaunchKernelGGL(k_frame_fast_t<0u>, dim3(grid), dim3(kBlock), 0, c->stream, d_img, nbytes, n_chunks,
                       (const uint32_t*)c->tile_local.p, (const unsigned long long*)c->span_sums.p,
                       (uint64_t*)c->line_end.p, line_cap, 4 * (usable / 4), sm, c->d_cs, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr);
  else
    hipLaunchKernelGGL(k_frame_fast_t<7u>, dim3(grid), dim3(kBlock), 0, c->stream, d_img, nbytes, n_chunks,
                       (const uint32_t*)c->tile_local.p, (const unsigned long long*)c->span_sums.p,
                       (uint64_t*)c->line_end.p, line_cap, (uint64_t)0, sm, c->d_cs, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr);
  out->checks_done = checks;
  return 0;
}


constexpr uint32_t kStreamBootBytes = kBootMax;   // prefix whose quality range seeds the range test (64 KiB: LDS of k_stream_boot)
constexpr uint64_t kStreamQueueCap = 1ull << 20;

// One pass over the image (see fqg_stream_kernels.hip).  Returns 1 when the image is not eligible
// (NUL / CR bytes, bytes >= 0x80, more newlines per chunk than the staging area holds): the caller
// then runs frame_two_pass, which also decides about the exact path.
struct RecordDuties {  // what k_stream_lines needs from the caller of frame_stream
  int space;
  uint32_t weight;
  AccState* acc;
  unsigned long long* hist;
  int fmt, is_pe;  // name digests are made under the caller's file state
};

int frame_stream(fqg_ctx* c, const uint8_t* d_img, uint64_t nbytes, uint32_t n_chunks, bool final, SuspectMap sm,
                 const RecordDuties& rd, int want_names /* 0, 1 = records, 2 = digests */, bool want_index, Framed* out) {
  int rc;
  c->names_img = nullptr;
  const uint32_t n_spans = (n_chunks + kScanSpan - 1) / kScanSpan;
  if ((rc = ensure(c, c->tile_counts, (size_t)n_chunks * 4))) return rc;
  if ((rc = ensure(c, c->tile_local, (size_t)n_chunks * 4))) return rc;
  if ((rc = ensure(c, c->span_sums, (size_t)n_spans * 8))) return rc;
  if ((rc = ensure(c, c->cinfo, (size_t)n_chunks * 4))) return rc;
  if ((rc = ensure(c, c->redo, (size_t)n_chunks * 4))) return rc;
  if ((rc = ensure(c, c->stage, (size_t)n_chunks * kStageCap * 2))) return rc;
  if ((rc = ensure(c, c->queue, (size_t)kStreamQueueCap * 8))) return rc;
  init_call_state(c);
  HIP_TRY(c, hipMemcpyAsync(c->d_cs, c->h_cs, sizeof(CallState), hipMemcpyHostToDevice, c->stream));
  StreamOut so;
  so.counts = (uint32_t*)c->tile_counts.p;
  so.cinfo = (uint32_t*)c->cinfo.p;
  so.stage = (uint16_t*)c->stage.p;
  so.queue = (unsigned long long*)c->queue.p;
  so.queue_cap = kStreamQueueCap;
  {
    ProfScope ps(c, "k_stream_boot");
    const uint32_t span = (uint32_t)std::min<uint64_t>(kStreamBootBytes, nbytes & ~255ull);
    hipLaunchKernelGGL(k_stream_boot, dim3(1), dim3(kBootBlock), 0, c->stream, d_img, span, c->d_cs);
  }
  NameCapture nc{};
  if (want_names) {
    // record slots per chunk from the mean record of the boot window: a power of two with a quarter of headroom
    // (a chunk that sees more headers is read through the line index instead)
    HIP_TRY(c, hipMemcpyAsync(c->h_cs, c->d_cs, sizeof(CallState), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint32_t span = (uint32_t)std::min<uint64_t>(kStreamBootBytes, nbytes & ~255ull);
    const double per_chunk = (double)(c->h_cs->boot_lines / 4 + 1) * (double)kChunkBytes / (double)std::max<uint32_t>(span, 1);
    uint32_t shift = 3;
    while (shift < 7 && (double)(1u << shift) < 1.25 * per_chunk + 1.0) ++shift;
    nc.K = 1u << shift;
    nc.fmt = rd.fmt;
    nc.is_pe = rd.is_pe;
    if ((rc = ensure(c, c->name_recs, ((size_t)n_chunks << shift) * (want_names == 2 ? kDigestWords : kNameRecWords) * 8))) return rc;
    if ((rc = ensure(c, c->name_hcount, (size_t)n_chunks * 2))) return rc;
    nc.recs = (unsigned long long*)c->name_recs.p;
    nc.hcount = (uint16_t*)c->name_hcount.p;
    c->names.k_shift = shift;
  }
  // Large images that only want to be validated take the pass in PARTS: pass 1 of part p shares its launch with the line
  // workers of part p - 1 (k_stream_pass1_lines), whose statistics wait in accumulators of the context until the whole
  // image has passed without a flag that sends it to the two-pass path.  FQGPU_STREAM_PARTS=1: one launch, as before.
  // (read per call: tests and A/B runs switch them inside one process)
  const int parts_env = env_int_early("FQGPU_STREAM_PARTS", 4);
  const int workers_env = env_int_early("FQGPU_STREAM_LINE_WORKERS", 1);  // line-worker workgroups per CU
  const uint32_t parts_min_spans = (uint32_t)std::max(env_int_early("FQGPU_STREAM_PARTS_MIN_SPANS", 24), 1);  // (16 MiB each)
  static const bool old_pass2_early = getenv("FQGPU_STREAM_PASS2_OLD") != nullptr;
  // (with the index wanted - the name modes - the line workers store it as they go, into room sized from the boot window)
  uint32_t n_parts = ((want_names || !want_index) && !old_pass2_early && rd.acc && n_spans >= parts_min_spans)
                         ? (uint32_t)std::min(std::max(parts_env, 1), 4) : 1u;
  n_parts = std::min(n_parts, n_spans);
  uint64_t early_line_cap = 0;
  if (n_parts > 1 && want_index) {
    // lines of the image from the lines of the boot window (the name modes have waited for it above), a quarter more
    const uint32_t span = (uint32_t)std::min<uint64_t>(kStreamBootBytes, nbytes & ~255ull);
    const double per_byte = (double)(c->h_cs->boot_lines + 4) / (double)std::max<uint32_t>(span, 1);
    early_line_cap = (uint64_t)(per_byte * 1.25 * (double)nbytes) + 4096;
    if ((rc = ensure(c, c->line_end, (size_t)early_line_cap * 8))) return rc;
  }
  LinesArgs PA{};  // what the line workers inside the pass-1 launches go by (the rest comes from the call state)
  uint64_t todo_steps_cap = 0;
  if (n_parts > 1) {
    if (!c->pipe_hist) {
      if (!c->pipe_acc) HIP_TRY(c, hipMalloc((void**)&c->pipe_acc, sizeof(AccState)));
      HIP_TRY(c, hipMalloc((void**)&c->pipe_hist, sizeof(unsigned long long) * FQG_MAX_READ_LENGTH));
      HIP_TRY(c, hipMemsetAsync(c->pipe_hist, 0, sizeof(unsigned long long) * FQG_MAX_READ_LENGTH, c->stream));
      hipLaunchKernelGGL(k_acc_scratch_reset, dim3(1), dim3(1), 0, c->stream, c->pipe_acc);
      c->pipe_dirty = false;
    }
    if (c->pipe_dirty) {
      hipLaunchKernelGGL(k_acc_merge, dim3(64), dim3(kBlock), 0, c->stream, c->pipe_acc, c->pipe_hist, (AccState*)nullptr,
                         (unsigned long long*)nullptr, 1);
      hipLaunchKernelGGL(k_acc_scratch_reset, dim3(1), dim3(1), 0, c->stream, c->pipe_acc);
    }
    c->pipe_dirty = true;
    // one mark per step for the general line kernel: every line has a byte, 512 lines a step
    todo_steps_cap = nbytes / (uint64_t)(4 * kWave * kLinesPer) + 8;
    if ((rc = ensure(c, c->lines_slow, (size_t)todo_steps_cap + 4))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->lines_slow.p, 0, (size_t)todo_steps_cap, c->stream));
    PA.img = d_img;
    PA.n = nbytes;
    PA.cr.counts = (const uint32_t*)c->tile_counts.p;
    PA.cr.local = (const uint32_t*)c->tile_local.p;
    PA.cr.span_excl = (const unsigned long long*)c->span_sums.p;
    PA.stage = (const uint16_t*)c->stage.p;
    PA.line_end = want_index ? (uint64_t*)c->line_end.p : nullptr;
    PA.line_cap = want_index ? early_line_cap : ~0ull;
    PA.suspect_bits = sm.bits;
    PA.suspect_cap = sm.cap;
    PA.flags = sm.flags;
    PA.space = rd.space;
    PA.weight = rd.weight;
    PA.acc = c->pipe_acc;
    PA.hist = c->pipe_hist;
    PA.ablate = 0;
    PA.no_index = want_index ? 0u : 1u;
    PA.index_only = 0u;
    PA.keep_from = ~0ull;
  }
  // The LAST part is the small one: its line work has no pass-1 launch behind it to hide in - it is what runs alone
  // at the end of the call - while the line work of the parts before it runs inside the next part's launch at next to no
  // cost.  FQGPU_STREAM_LAST_PART_PCT (default 16: parts of 28 / 28 / 28 / 16 % of the image; 25: equal parts, as this was
  // first built).  The line workers are not quite free - what the tail loses the launches gain back in part: one box, two
  // runs each, ms per step: 25 % 7.89 / 7.96, 16 % 7.88 / 7.87, 12 % 7.96 / 7.90, 8 % 7.96 / 7.98, 5 % 8.19 / 8.20 (there
  // pass 1 of the last part no longer outlasts the line workers of the part before).
  // (The name modes, whose line workers store the line index as they go, like it too - the pass with digests, one box, two
  // runs each: 25 % 10.10 / 10.12 ms + 0.34 behind it, 16 % 10.01 / 10.06 + 0.26.  Boxes differ by more than that.)
  const int last_pct = std::min(std::max(env_int_early("FQGPU_STREAM_LAST_PART_PCT", 16), 1), 100);
  auto part_begin = [&](uint32_t part) -> uint32_t {
    if (part == 0 || n_parts <= 1) return part == 0 ? 0u : n_spans;
    if (part >= n_parts) return n_spans;
    const uint32_t last = std::min<uint32_t>(n_spans - (n_parts - 1), std::max<uint32_t>(1u, (uint32_t)((uint64_t)n_spans * (uint32_t)last_pct / 100u)));
    return (uint32_t)((uint64_t)(n_spans - last) * part / (n_parts - 1));
  };
  for (uint32_t part = 0; part < n_parts; ++part) {
    const uint32_t span_lo = part_begin(part), span_hi = part_begin(part + 1);
    const uint32_t chunk_lo = span_lo * kScanSpan, chunk_hi = std::min<uint32_t>(n_chunks, span_hi * kScanSpan);
    {
      ProfScope ps(c, want_names == 2 ? (part ? "k_stream_pass1_lines(digests)" : "k_stream_pass1(digests)")
                      : want_names    ? (part ? "k_stream_pass1_lines(names)" : "k_stream_pass1(names)")
                      : part          ? "k_stream_pass1_lines" : "k_stream_pass1");
      const unsigned blocks = (chunk_hi - chunk_lo + 3) / 4;
      const unsigned workers = (unsigned)c->cu_count * (unsigned)std::min(std::max(workers_env, 1), 4);
#define FQG_PASS1_PART(N)                                                                                                    \
  do {                                                                                                                       \
    if (part == 0)                                                                                                           \
      hipLaunchKernelGGL((k_stream_pass1<0u, N>), dim3(blocks), dim3(kBlock), 0, c->stream, d_img, nbytes, chunk_hi, so,      \
                         c->d_cs, nc);                                                                                       \
    else                                                                                                                     \
      hipLaunchKernelGGL(k_stream_pass1_lines<N>, dim3(workers + blocks), dim3(kBlock), 0, c->stream, d_img, nbytes, chunk_lo, \
                         chunk_hi, so, c->d_cs, PA, (uint8_t*)c->lines_slow.p, workers, part - 1, nc);                        \
  } while (0)
      if (want_names == 2) FQG_PASS1_PART(2);
      else if (want_names) FQG_PASS1_PART(1);
      else FQG_PASS1_PART(0);
#undef FQG_PASS1_PART
    }
    {
      ProfScope ps(c, "k_scan");
      // (two launches: ONE, with the workgroup that finishes last doing phase B, was tried - the release fence every
      // workgroup needs in front of its count writes the L2 back: 0.21 ms a step instead of 0.11)
      hipLaunchKernelGGL(k_scan_a, dim3(span_hi - span_lo), dim3(kBlock), 0, c->stream, (const uint32_t*)c->tile_counts.p,
                         n_chunks, (uint32_t*)c->tile_local.p, (unsigned long long*)c->span_sums.p, span_lo);
      hipLaunchKernelGGL(k_scan_b, dim3(1), dim3(kBlock), 0, c->stream, (unsigned long long*)c->span_sums.p,
                         span_hi, d_img, nbytes, c->d_cs, span_lo, part);
    }
  }
  // (leaving the parted pass without its statistics: the scratch is cleared, nothing reaches the caller's accumulator)
  auto drop_parted = [&]() {
    if (n_parts > 1) {
      hipLaunchKernelGGL(k_acc_merge, dim3(64), dim3(kBlock), 0, c->stream, c->pipe_acc, c->pipe_hist, (AccState*)nullptr,
                         (unsigned long long*)nullptr, 1);
      hipLaunchKernelGGL(k_acc_scratch_reset, dim3(1), dim3(1), 0, c->stream, c->pipe_acc);
      c->pipe_dirty = false;
    }
  };
  HIP_TRY(c, hipMemcpyAsync(c->h_cs, c->d_cs, sizeof(CallState), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->h_cs->flags & (kFlagNul | kFlagCr | kFlagHigh | kFlagStageOverflow)) {
    drop_parted();
    return 1;
  }
  out->n_newlines = c->h_cs->n_newlines;
  out->last_nl = c->h_cs->last_byte_is_nl != 0;
  out->img_flags = 0;
  const uint64_t n_lines_all = out->n_newlines + (out->last_nl ? 0 : 1);
  const uint64_t usable = (final || out->last_nl) ? n_lines_all : out->n_newlines;
  const uint64_t line_cap = n_lines_all + 17;
  const uint64_t limit = 4 * (usable / 4);
  // (the parted pass of a name mode: its line workers have stored index entries already - into room sized from the
  // boot window.  Too small after all: a new allocation, and their stores once more, below)
  const bool index_lost = early_line_cap && (size_t)line_cap * 8 > c->line_end.cap;
  if ((rc = ensure(c, c->line_end, (size_t)line_cap * 8))) return rc;
  static const bool old_pass2 = getenv("FQGPU_STREAM_PASS2_OLD") != nullptr;  // (A/B: the chunk-owned second pass)
  if (old_pass2) {
    ProfScope ps(c, "k_stream_pass2");
    const unsigned per_wg = (kBlock / kWave) * kP2Batch;
    hipLaunchKernelGGL(k_stream_pass2, dim3((n_chunks + per_wg - 1) / per_wg), dim3(kBlock), 0, c->stream, d_img, nbytes, n_chunks,
                       (const uint32_t*)c->tile_counts.p, (const uint32_t*)c->cinfo.p, (const uint16_t*)c->stage.p,
                       (const uint32_t*)c->tile_local.p, (const unsigned long long*)c->span_sums.p,
                       (uint64_t*)c->line_end.p, line_cap, limit, sm, (uint32_t*)c->redo.p, c->d_cs);
  } else {
    ChunkRanks cr;
    cr.counts = (const uint32_t*)c->tile_counts.p;
    cr.local = (const uint32_t*)c->tile_local.p;
    cr.span_excl = (const unsigned long long*)c->span_sums.p;
    cr.n_chunks = n_chunks;
    {
      ProfScope ps(c, "k_stream_chunks");
      hipLaunchKernelGGL(k_stream_chunks, dim3((n_chunks + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, cr,
                         (const uint32_t*)c->cinfo.p, limit, (uint32_t*)c->redo.p, c->d_cs);
    }
    LinesArgs A;
    A.img = d_img;
    A.n = nbytes;
    A.cr = cr;
    A.stage = (const uint16_t*)c->stage.p;
    A.line_end = (uint64_t*)c->line_end.p;
    A.line_cap = line_cap;
    A.n_newlines = out->n_newlines;
    A.n_lines = n_lines_all;
    A.limit = limit;
    A.suspect_bits = sm.bits;
    A.suspect_cap = sm.cap;
    A.flags = sm.flags;
    A.space = rd.space;
    A.weight = rd.weight;
    A.acc = rd.acc;
    A.hist = rd.hist;
    static const int lines_abl = measure_int("FQGPU_LINES_ABL");
    A.ablate = lines_abl;
    // the index on demand: when the caller has not said it wants it, and nothing queued by pass 1 needs it (the
    // queue kernel finds its records through the index).  FQGPU_EAGER_INDEX=1: always in this call (A/B)
    static const bool eager_index = getenv("FQGPU_EAGER_INDEX") != nullptr;
    const bool lazy = !want_index && !eager_index && c->h_cs->queue_count == 0;
    A.no_index = lazy ? 1u : 0u;
    A.index_only = 0u;
    A.keep_from = limit >= 8 ? limit - 8 : 0;
    c->lazy.pending = false;
    // the steps the line workers inside the pass-1 launches have done (the parted pass): the launches below begin behind
    // them, and every line kernel of such a call adds to the context's accumulators, merged at the end
    const uint64_t done_steps = n_parts > 1 ? c->h_cs->lines_done_steps : 0;
    if (n_parts > 1) {
      A.acc = c->pipe_acc;
      A.hist = c->pipe_hist;
    }
    {
      ProfScope ps(c, "k_stream_lines");
      const uint64_t groups = (n_lines_all + 4 * kWave - 1) / (4 * kWave);
      // a persistent grid: every workgroup must be resident from the start (one that is not would do its whole
      // share after the others have finished)
      if (!c->lines_per_cu) {  // (per context: a context is one device, and contexts run on threads of their own)
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_stream_lines<false>), kBlock, 0) != hipSuccess || nb < 1)
          nb = 4;
        c->lines_per_cu = nb;
      }
      const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((groups + 3) / 4, (uint64_t)c->cu_count * c->lines_per_cu));
      static const bool general_only = getenv("FQGPU_LINES_GENERAL") != nullptr;  // (A/B: every step through the general kernel)
      // (more than 64 newlines in an average chunk - reads below 100 bases - or fewer than 32 - reads of kilobases: hardly a
      // step would qualify for the kernel without a search)
      const double nl_per_chunk = (double)out->n_newlines / (double)std::max<uint32_t>(n_chunks, 1);
      c->lazy.grid = grid;
      c->lazy.fast = false;
      if (done_steps == 0 && (general_only || nl_per_chunk > 60.0 || nl_per_chunk < 32.0)) {
        hipLaunchKernelGGL(k_stream_lines<false>, dim3(grid), dim3(kBlock), 0, c->stream, A, (const uint8_t*)nullptr);
      } else {
        // the steps of ordinary records in the kernel without a search, what it marks in the general one behind it
        const uint64_t n_steps = (groups + kLinesPer - 1) / kLinesPer;
        if (n_parts > 1 && n_steps + 4 > todo_steps_cap) return fail(c, FQG_ERR_STATE, "streaming pass: more steps than lines");
        if (n_parts == 1) {
          if ((rc = ensure(c, c->lines_slow, (size_t)n_steps + 4))) return rc;
          HIP_TRY(c, hipMemsetAsync(c->lines_slow.p, 0, (size_t)n_steps, c->stream));
        }
        if (!c->lines_fast_per_cu) {
          int nb = 0;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k_stream_lines_fast), kBlock, 0) != hipSuccess || nb < 1)
            nb = 4;
          c->lines_fast_per_cu = nb;
        }
        const unsigned grid_f = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_steps - std::min(done_steps, n_steps) + 3) / 4, (uint64_t)c->cu_count * c->lines_fast_per_cu));
        if (done_steps && !lazy && (!early_line_cap || index_lost)) {
          // the index is wanted after all (pass 1 queued byte positions), or its room was too small: the stores of the
          // steps that are done
          LinesArgs I = A;
          I.index_only = 1u;
          I.acc = nullptr;
          I.hist = nullptr;
          I.step_lo = 0;
          I.step_hi = done_steps;
          const unsigned grid_i = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((done_steps + 3) / 4, (uint64_t)c->cu_count * c->lines_fast_per_cu));
          hipLaunchKernelGGL(k_stream_lines_fast, dim3(grid_i), dim3(kBlock), 0, c->stream, I, (uint8_t*)c->lines_slow.p);
        }
        LinesArgs F = A;
        F.step_lo = done_steps;
        F.step_hi = 0;
        hipLaunchKernelGGL(k_stream_lines_fast, dim3(grid_f), dim3(kBlock), 0, c->stream, F, (uint8_t*)c->lines_slow.p);
        hipLaunchKernelGGL(k_stream_lines<true>, dim3(grid), dim3(kBlock), 0, c->stream, A, (const uint8_t*)c->lines_slow.p);
        c->lazy.grid_f = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_steps + 3) / 4, (uint64_t)c->cu_count * c->lines_fast_per_cu));
        c->lazy.fast = true;
      }
    }
    if (n_parts > 1) {
      ProfScope ps(c, "k_acc_merge");
      hipLaunchKernelGGL(k_acc_merge, dim3(64), dim3(kBlock), 0, c->stream, c->pipe_acc, c->pipe_hist, rd.acc, rd.hist, 0);
      hipLaunchKernelGGL(k_acc_scratch_reset, dim3(1), dim3(1), 0, c->stream, c->pipe_acc);
      c->pipe_dirty = false;
      A.acc = rd.acc;
      A.hist = rd.hist;
    }
    if (lazy) {
      c->lazy.pending = true;
      c->lazy.args = A;
      c->lazy.args.no_index = 0u;
      c->lazy.args.index_only = 1u;
      c->lazy.args.acc = nullptr;
      c->lazy.args.hist = nullptr;
    }
    out->records_done = true;
  }
  if (!c->lazy.pending) {  // (on demand only when pass 1 queued nothing: see above)
    ProfScope ps(c, "k_stream_queue");
    hipLaunchKernelGGL(k_stream_queue, dim3(64), dim3(kBlock), 0, c->stream, (const unsigned long long*)c->queue.p,
                       (unsigned long long)kStreamQueueCap, (const uint64_t*)c->line_end.p, n_lines_all, limit / 4, sm,
                       c->d_cs);
  }
  {
    // chunks whose speculated line type was wrong or missing: the two-pass kernel repeats the checks
    // with the true rank (no line-index stores)
    ProfScope ps(c, "k_stream_redo");
    // (ordinary reads send a handful of chunks here - the ones whose speculation failed -, reads of kilobases a quarter of
    // all: a grid that fills the GPU costs 0.06 ms to start and end for the former)
    const double nlpc = (double)out->n_newlines / (double)std::max<uint32_t>(n_chunks, 1);
    const unsigned grid = (unsigned)std::min<uint64_t>((n_chunks + 3) / 4, (uint64_t)c->cu_count * (nlpc >= 32.0 ? 1 : 8));
    hipLaunchKernelGGL(k_frame_fast_t<8u>, dim3(grid), dim3(kBlock), 0, c->stream, d_img, nbytes, n_chunks,
                       (const uint32_t*)c->tile_local.p, (const unsigned long long*)c->span_sums.p,
                       (uint64_t*)c->line_end.p, line_cap, limit, sm, c->d_cs, (const uint32_t*)c->redo.p,
                       (const uint32_t*)&c->d_cs->redo_count);
  }
  out->checks_done = true;
  if (want_names && !old_pass2) {
    c->names.recs = nc.recs;
    c->names.hcount = nc.hcount;
    c->names.cinfo = (const uint32_t*)c->cinfo.p;
    c->names.cr.counts = (const uint32_t*)c->tile_counts.p;
    c->names.cr.local = (const uint32_t*)c->tile_local.p;
    c->names.cr.span_excl = (const unsigned long long*)c->span_sums.p;
    c->names.cr.n_chunks = n_chunks;
    c->names.K = nc.K;
    c->names.digests = want_names == 2 ? 1u : 0u;
    c->names.fmt = rd.fmt;
    c->names.is_pe = rd.is_pe;
    c->names_img = d_img;
    c->names_nbytes = nbytes;
  }
  return 0;
}

// The line index of the current frame, if the call that framed it left it for later (LinesArgs::no_index): the line
// kernels once more, stores only, on the context's stream - whatever is launched behind them there finds it written.
int index_now(fqg_ctx* c) {
  if (!c->lazy.pending) return 0;
  c->lazy.pending = false;
  HIP_TRY(c, hipSetDevice(c->device));
  ProfScope ps(c, "k_stream_lines_index");
  const LinesArgs& A = c->lazy.args;
  if (c->lazy.fast) {
    hipLaunchKernelGGL(k_stream_lines_fast, dim3(c->lazy.grid_f), dim3(kBlock), 0, c->stream, A, (uint8_t*)c->lines_slow.p);
    hipLaunchKernelGGL(k_stream_lines<true>, dim3(c->lazy.grid), dim3(kBlock), 0, c->stream, A, (const uint8_t*)c->lines_slow.p);
  } else {
    hipLaunchKernelGGL(k_stream_lines<false>, dim3(c->lazy.grid), dim3(kBlock), 0, c->stream, A, (const uint8_t*)nullptr);
  }
  HIP_TRY(c, hipGetLastError());
  return 0;
}

}  // namespace

int fqg_validate(fqg_ctx* c, fqg_acc* acc, const void* image, uint64_t nbytes, int mem, int final,
                 const fqg_file_state* st, uint32_t flags, fqg_validate_result* out) {
  if (!c || !out || !st || (nbytes && !image)) return FQG_ERR_ARG;
  if (!(flags & (FQG_VALIDATE_NO_STATS | FQG_VALIDATE_FRAME_ONLY)) && !acc)
    return fail(c, FQG_ERR_ARG, "fqg_validate: acc is NULL");
  if (mem != FQG_MEM_HOST && mem != FQG_MEM_DEVICE) return FQG_ERR_ARG;
  if (flags & FQG_VALIDATE_NO_STATS) acc = nullptr;
  memset(out, 0, sizeof(*out));
  c->frame_valid = false;
  c->frame_borrowed = false;
  c->lazy.pending = false;
  c->names_img = nullptr;
  HIP_TRY(c, hipSetDevice(c->device));
  if (nbytes == 0) return 0;
  if (nbytes >= (1ull << 44)) return fail(c, FQG_ERR_ARG, "image too large");

  int rc;
  const uint8_t* d_img;
  if (mem == FQG_MEM_HOST) {
    if ((rc = ensure(c, c->image, nbytes + 64))) return rc;
    ProfScope ps(c, "h2d_image");
    HIP_TRY(c, hipMemcpyAsync(c->image.p, image, nbytes, hipMemcpyHostToDevice, c->stream));
    d_img = (const uint8_t*)c->image.p;
  } else {
    if (((uintptr_t)image & 15u) != 0) return fail(c, FQG_ERR_ARG, "device image must be 16-byte aligned");
    d_img = (const uint8_t*)image;
  }
  const uint32_t n_chunks = (uint32_t)((nbytes + kChunkBytes - 1) / kChunkBytes);
  const bool frame_only = (flags & FQG_VALIDATE_FRAME_ONLY) != 0;
  if (frame_only) acc = nullptr;
  // (frame only + names: the single-pass framing runs for its capture records; what its checks find is not looked at)
  const bool names_only = frame_only && (flags & (FQG_VALIDATE_NAMES | FQG_VALIDATE_NAME_DIGESTS)) && nbytes >= c->stream_min && !(flags & FQG_VALIDATE_TWO_PASS);
  const bool want_checks = !(flags & FQG_VALIDATE_FORCE_EXACT) && (!frame_only || names_only);
  const uint32_t weight = (flags & FQG_VALIDATE_COUNT_TWICE) ? 2u : 1u;

  // suspect bitmap, sized from the image (>= 16 bytes per record assumed; denser images overflow
  // it, which sends every record to the exact validator)
  SuspectMap sm;
  sm.cap = std::max<uint64_t>(nbytes / 16, 1024);
  sm.flags = &c->d_cs->flags;
  sm.bits = nullptr;
  if (want_checks) {
    if ((rc = ensure(c, c->suspect, (size_t)(sm.cap / 32 + 2) * 4))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->suspect.p, 0, (size_t)(sm.cap / 32 + 2) * 4, c->stream));
    sm.bits = (uint32_t*)c->suspect.p;
  }

  Framed fr;
  bool streamed = false;
  if (want_checks && nbytes >= c->stream_min && !(flags & FQG_VALIDATE_TWO_PASS)) {
    RecordDuties rd{st->space, weight, acc ? acc->d_state : nullptr, acc ? acc->d_hist : nullptr, st->readname_format, st->is_pe};
    // (digests need a format to canonicalise under: a file state that has none yet gets records)
    const int names_mode = (flags & FQG_VALIDATE_NAME_DIGESTS) && st->readname_format != FQG_NAME_UNDEF ? 2
                           : (flags & (FQG_VALIDATE_NAMES | FQG_VALIDATE_NAME_DIGESTS)) ? 1 : 0;
    rc = frame_stream(c, d_img, nbytes, n_chunks, final != 0, sm, rd, names_mode,
                      (flags & (FQG_VALIDATE_NAMES | FQG_VALIDATE_NAME_DIGESTS | FQG_VALIDATE_INDEX)) != 0, &fr);
    if (rc < 0) return rc;
    streamed = rc == 0;
    if (!streamed) HIP_TRY(c, hipMemsetAsync(c->suspect.p, 0, (size_t)(sm.cap / 32 + 2) * 4, c->stream));
  }
  if (!streamed && (rc = frame_two_pass(c, d_img, nbytes, n_chunks, final != 0, want_checks && !frame_only, sm, &fr))) return rc;
  const uint64_t n_newlines = fr.n_newlines;
  const bool last_nl = fr.last_nl;
  const uint32_t img_flags = fr.img_flags;
  const uint64_t n_lines_all = n_newlines + (last_nl ? 0 : 1);
  // an unterminated last line is only a line when nothing more can follow
  const uint64_t usable = (final || last_nl) ? n_lines_all : n_newlines;
  uint64_t n_records = usable / 4;
  const uint64_t leftover = usable % 4;

  FrameView fv;
  fv.img = d_img;
  fv.nbytes = nbytes;
  fv.line_end = (const uint64_t*)c->line_end.p;
  fv.n_lines = n_lines_all;
  fv.n_records = n_records;
  fv.reframed = (flags & FQG_VALIDATE_REFRAMED) ? 1u : 0u;
  if (!fv.reframed) {
    // the gzgets limits, where the record-wise checks do not look: every line of an image that is only framed, and the
    // lines of an incomplete last record (the reference reads THOSE in pieces too before it finds the file truncated)
    const uint64_t from = frame_only ? 0 : 4 * n_records;
    const uint64_t upto = final ? n_lines_all : usable;
    if (upto > from) {
      ProfScope ps(c, "k_overlong");
      FrameView lv = fv;
      lv.n_lines = upto;
      hipLaunchKernelGGL(k_overlong, dim3((unsigned)((upto - from + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, lv, from,
                         c->d_cs);
    }
  }

  // a record that starts with NUL ends the file silently (src/fastq.c:250)
  bool tail_is_stop = false, nul_truncated = false;
  if (img_flags & kFlagNul) {
    if (n_records) {
      ProfScope ps(c, "k_find_stop");
      hipLaunchKernelGGL(k_find_stop, dim3((unsigned)((n_records + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                         c->stream, fv, c->d_cs, frame_only ? 1 : 0);
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_cs, c->d_cs, sizeof(CallState), hipMemcpyDeviceToHost, c->stream));
    if (leftover && final) {
      // first byte of the incomplete trailing group
      const uint64_t* le = (const uint64_t*)c->line_end.p;
      if (n_records) {
        HIP_TRY(c, hipMemcpyAsync(&c->h_scalar[0], le + 4 * n_records - 1, 8, hipMemcpyDeviceToHost, c->stream));
      } else c->h_scalar[0] = ~0ull;
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      const uint64_t at = c->h_scalar[0] + 1;
      uint8_t b = 1;
      HIP_TRY(c, hipMemcpy(&b, d_img + at, 1, hipMemcpyDeviceToHost));
      tail_is_stop = (b == 0);
    } else {
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (c->h_cs->stop_record < n_records) {
      n_records = c->h_cs->stop_record;
      out->stopped = 1;
    }
    if (c->h_cs->trunc_record < n_records) {  // (frame-only: an earlier record with an empty line - the file is truncated THERE)
      n_records = c->h_cs->trunc_record;
      out->stopped = 0;
      nul_truncated = true;
    }
    fv.n_records = n_records;
  }

  // The tiled path needs an image without NUL / CR bytes (then every line is terminated by its
  // '\n' alone and the statistics follow from the line index); anything else goes through the
  // exact wave-per-record validator as a whole.
  const bool fast = fr.checks_done && !frame_only;
  uint64_t list_cap = 0;
  if (fast) {
    out->path = streamed ? 3 : 2;
    if (n_records) {
      list_cap = std::max<uint64_t>(1u << 20, n_records / 16);
      if (list_cap > n_records) list_cap = n_records;
      if ((rc = ensure(c, c->list, (size_t)list_cap * 8))) return rc;
      const unsigned grid_r =
          (unsigned)std::min<uint64_t>((n_records + kBlock - 1) / kBlock, (uint64_t)c->cu_count * 8);
      if (fr.records_done) {
        ProfScope ps(c, "k_suspect_list");
        hipLaunchKernelGGL(k_suspect_list, dim3((unsigned)std::min<uint64_t>((n_records / 32 + kBlock) / kBlock, 1024)), dim3(kBlock), 0,
                           c->stream, (const uint32_t*)sm.bits, std::min<uint64_t>(n_records, sm.cap),
                           (unsigned long long*)c->list.p, list_cap, &c->d_cs->list_count, acc ? acc->d_state : nullptr,
                           (const CallState*)c->d_cs);
      } else {
        ProfScope ps(c, "k_records_fast");
        hipLaunchKernelGGL(k_records_fast, dim3(grid_r), dim3(kBlock), 0, c->stream, fv, st->space, weight, sm,
                           (unsigned long long*)c->list.p, list_cap, &c->d_cs->list_count,
                           acc ? acc->d_state : nullptr, acc ? acc->d_hist : nullptr, c->d_cs);
      }
      if (!c->lazy.pending) {  // (on demand: once the number of listed records is known, below)
        ProfScope ps(c, "k_validate_exact");
        hipLaunchKernelGGL(k_validate_exact, dim3(c->cu_count * 2), dim3(kBlock), 0, c->stream, fv, st->is_pe,
                           st->readname_format, st->space, weight, (AccState*)nullptr,
                           (unsigned long long*)nullptr, c->d_cs, kNoRecord,
                           (const unsigned long long*)c->list.p, (const unsigned long long*)&c->d_cs->list_count);
      }
    }
  } else if (n_records && !frame_only) {
    ProfScope ps(c, "k_validate_exact");
    hipLaunchKernelGGL(k_validate_exact, dim3(grid_for_waves(c, n_records)), dim3(kBlock), 0, c->stream, fv,
                       st->is_pe, st->readname_format, st->space, weight, acc ? acc->d_state : nullptr,
                       acc ? acc->d_hist : nullptr, c->d_cs, kNoRecord, (const unsigned long long*)nullptr,
                       (const unsigned long long*)nullptr);
    out->path = 1;
  }
  HIP_TRY(c, hipMemcpyAsync(c->h_cs, c->d_cs, sizeof(CallState), hipMemcpyDeviceToHost, c->stream));
  if (n_records) {
    HIP_TRY(c, hipMemcpyAsync(&c->h_scalar[1], (const uint64_t*)c->line_end.p + 4 * n_records - 1, 8,
                              hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->lazy.pending && fast && n_records && c->h_cs->list_count > 0 && c->h_cs->list_count <= list_cap &&
      !(c->h_cs->flags & (kFlagSuspectOverflow | kFlagQueueOverflow))) {
    // records the line kernels could not vouch for: the exact validator reads them through the index
    if ((rc = index_now(c))) return rc;
    ProfScope ps(c, "k_validate_exact");
    hipLaunchKernelGGL(k_validate_exact, dim3(c->cu_count * 2), dim3(kBlock), 0, c->stream, fv, st->is_pe,
                       st->readname_format, st->space, weight, (AccState*)nullptr,
                       (unsigned long long*)nullptr, c->d_cs, kNoRecord,
                       (const unsigned long long*)c->list.p, (const unsigned long long*)&c->d_cs->list_count);
    HIP_TRY(c, hipMemcpyAsync(c->h_cs, c->d_cs, sizeof(CallState), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  out->n_records = n_records;
  out->n_lines = n_lines_all;
  out->tail_lines = nul_truncated ? 1 : (final && leftover && !out->stopped && !tail_is_stop) ? (int32_t)leftover : 0;
  out->consumed = n_records ? c->h_scalar[1] + 1 : 0;
  if (out->consumed > nbytes) out->consumed = nbytes;  // unterminated last line

  if (fast && n_records &&
      (c->h_cs->list_count > list_cap || (c->h_cs->flags & (kFlagSuspectOverflow | kFlagQueueOverflow)))) {
    // more suspects than the queue / bitmap holds (e.g. every record carries its name on line 3):
    // let the exact validator look at every record; the statistics of the tiled pass stand
    if ((rc = index_now(c))) return rc;
    ProfScope ps(c, "k_validate_exact");
    hipLaunchKernelGGL(k_validate_exact, dim3(grid_for_waves(c, n_records)), dim3(kBlock), 0, c->stream, fv,
                       st->is_pe, st->readname_format, st->space, weight, (AccState*)nullptr,
                       (unsigned long long*)nullptr, c->d_cs, kNoRecord, (const unsigned long long*)nullptr,
                       (const unsigned long long*)nullptr);
    HIP_TRY(c, hipMemcpyAsync(c->h_cs, c->d_cs, sizeof(CallState), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (c->h_cs->first_key != ~0ull) {
    out->record = c->h_cs->first_key >> 8;
    out->code = (int32_t)(c->h_cs->first_key & 0xFF);
    if (out->code != FQG_E_LINE_TOO_LONG) {  // (that one has no arguments, and its record may be the incomplete last one)
      if ((rc = index_now(c))) return rc;
      hipLaunchKernelGGL(k_validate_exact, dim3(1), dim3(kBlock), 0, c->stream, fv, st->is_pe,
                         st->readname_format, st->space, 1u, (AccState*)nullptr, (unsigned long long*)nullptr,
                         c->d_cs, out->record, (const unsigned long long*)nullptr,
                         (const unsigned long long*)nullptr);
      HIP_TRY(c, hipMemcpyAsync(c->h_cs, c->d_cs, sizeof(CallState), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      out->aux0 = c->h_cs->aux0;
      out->aux1 = c->h_cs->aux1;
    }
  } else if (nul_truncated || (final && leftover && !out->stopped && !tail_is_stop)) {
    // src/fastq.c:254-257: fewer than four lines left (or, in an image that is only framed, a record with a line that
    // starts with NUL - an empty string to the reference: n_records counts the records in front of it)
    out->code = FQG_E_TRUNCATED;
    out->record = n_records;
  } else if (tail_is_stop && !out->stopped) {
    out->stopped = 1;
  }
  c->frame = fv;
  c->frame_valid = true;
  c->frame_img_owned = (mem == FQG_MEM_HOST);
  c->frame_flags = img_flags;
  HIP_TRY(c, hipGetLastError());
  return 0;
}

int fqg_frame_records(fqg_ctx* c, uint64_t first, uint64_t count, fqg_record* out, int mem) {
  if (!c || (!out && count)) return FQG_ERR_ARG;
  if (!c->frame_valid) return fail(c, FQG_ERR_STATE, "no frame: call fqg_validate first");
  if (const int irc = index_now(c)) return irc;
  if (first + count > c->frame.n_records) return fail(c, FQG_ERR_ARG, "record range outside the frame");
  if (!count) return 0;
  int rc;
  fqg_record* d_out = out;
  if (mem == FQG_MEM_HOST) {
    if ((rc = ensure(c, c->records, count * sizeof(fqg_record)))) return rc;
    d_out = (fqg_record*)c->records.p;
  }
  {
    ProfScope ps(c, "k_records");
    hipLaunchKernelGGL(k_records, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                       c->frame, first, count, d_out);
  }
  if (mem == FQG_MEM_HOST) {
    HIP_TRY(c, hipMemcpyAsync(out, d_out, count * sizeof(fqg_record), hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ---- frames ---------------------------------------------------------------------------------
struct fqg_frame {
  fqg_ctx* ctx = nullptr;
  FrameView fv{};
  DevBuf image;     // owned copy of a host image (empty for caller-owned device images)
  DevBuf line_end;  // owned line index
  uint32_t flags = 0;
};

int fqg_frame_make_current(fqg_ctx* c, const fqg_frame* f) {
  if (!c || !f || f->ctx != c) return FQG_ERR_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->frame = f->fv;
  c->lazy.pending = false;  // (a retained frame has its index: fqg_frame_retain)
  c->frame_flags = f->flags;
  c->frame_valid = true;
  c->frame_img_owned = false;
  c->frame_borrowed = true;
  return 0;
}

int fqg_frame_retain(fqg_ctx* c, fqg_frame** out) {
  if (!c || !out) return FQG_ERR_ARG;
  if (!c->frame_valid) return fail(c, FQG_ERR_STATE, "no frame: call fqg_validate first");
  if (const int irc = index_now(c)) return irc;
  if (c->frame_borrowed) return fail(c, FQG_ERR_STATE, "the current frame is a retained one: it cannot be retained again");
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  fqg_frame* f = new fqg_frame();
  f->ctx = c;
  f->fv = c->frame;
  f->flags = c->frame_flags;
  f->line_end = c->line_end;  // the context allocates fresh buffers on its next call
  c->line_end = DevBuf();
  if (c->frame_img_owned) {
    f->image = c->image;
    c->image = DevBuf();
  }
  c->frame_valid = false;
  *out = f;
  return 0;
}

void fqg_frame_release(fqg_frame* f) {
  if (!f) return;
  (void)hipStreamSynchronize(f->ctx->stream);
  release(f->image);
  release(f->line_end);
  delete f;
}

uint64_t fqg_frame_n_records(const fqg_frame* f) { return f ? f->fv.n_records : 0; }

// ---- read-name index ------------------------------------------------------------------------
struct fqg_index {
  fqg_ctx* ctx = nullptr;
  DevBuf slots, names, claims, segs_dev, found, match;
  uint64_t capacity = 0;
  uint64_t rec_cap = 0;          // records names[] / claims[] have room for
  bool keep_names = true;        // false: nobody will look names up in this index (fqg_index_expect_lookups)
  std::vector<fqg_frame*> frames;
  std::vector<IndexSeg> segs;
  uint64_t n_records_total = 0;  // records fed so far = global index of the next frame's first record
  uint64_t n_askers_total = 0;   // records of the asking file(s) matched against the index so far
  uint64_t inserted = 0, matched = 0, name_bytes = 0;
  int fmt = FQG_NAME_UNDEF, is_pe = 0;
  uint32_t flags = 0;
  bool names_once = true;        // no insert has reported a repeated name or a header without '@' (IndexView::n_positional)
};

namespace {

int index_alloc_table(fqg_index* ix, uint64_t capacity) {
  fqg_ctx* c = ix->ctx;
  int rc;
  if ((rc = ensure(c, ix->slots, (size_t)capacity * 8))) return rc;
  HIP_TRY(c, hipMemsetAsync(ix->slots.p, 0xFF, (size_t)capacity * 8, c->stream));
  ix->capacity = capacity;
  return 0;
}

IndexView index_view(fqg_index* ix) {
  IndexView v;
  v.slots = (unsigned long long*)ix->slots.p;
  v.names = ix->keep_names ? (NameRec*)ix->names.p : nullptr;
  v.claims = (unsigned long long*)ix->claims.p;
  v.mask = ix->capacity - 1;
  static const bool no_positional = getenv("FQGPU_NO_POSITIONAL_MATCH") != nullptr;  // (measurement, tests)
  v.n_positional = ix->keep_names && ix->names.p && ix->names_once && !no_positional ? ix->n_records_total : 0;
  v.segs = (const IndexSeg*)ix->segs_dev.p;
  v.n_segs = (int)ix->segs.size();
  v.fmt = ix->fmt;
  v.is_pe = ix->is_pe;
  v.may_have_nul = (ix->flags & kFlagNul) ? 1 : 0;
  return v;
}

// room in names[] / claims[] for `total` inserted records (kept across growth: record-ordered arrays)
int index_reserve_records(fqg_index* ix, uint64_t total) {
  fqg_ctx* c = ix->ctx;
  if (total <= ix->rec_cap) return 0;
  const uint64_t cap = std::max<uint64_t>(total + total / 8 + 1024, 2 * ix->rec_cap);
  DevBuf names, claims;
  hipError_t e = hipSuccess;
  if (ix->keep_names) e = hipMalloc(&names.p, (size_t)cap * sizeof(NameRec));
  if (e == hipSuccess) e = hipMalloc(&claims.p, (size_t)cap * 8);
  if (e != hipSuccess) {
    if (names.p) (void)hipFree(names.p);
    return fail(c, FQG_ERR_NOMEM, "name index: hipMalloc", e);
  }
  names.cap = (size_t)cap * sizeof(NameRec);
  claims.cap = (size_t)cap * 8;
  const uint64_t used = ix->n_records_total;
  if (used && ix->names.p && names.p)
    (void)hipMemcpyAsync(names.p, ix->names.p, (size_t)used * sizeof(NameRec), hipMemcpyDeviceToDevice, c->stream);
  if (used && ix->claims.p) (void)hipMemcpyAsync(claims.p, ix->claims.p, (size_t)used * 8, hipMemcpyDeviceToDevice, c->stream);
  (void)hipMemsetAsync((char*)claims.p + (size_t)used * 8, 0xFF, (size_t)(cap - used) * 8, c->stream);  // nobody has asked yet
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  release(ix->names);
  release(ix->claims);
  ix->names = names;
  ix->claims = claims;
  ix->rec_cap = cap;
  return 0;
}

int index_upload_segs(fqg_index* ix) {
  fqg_ctx* c = ix->ctx;
  int rc;
  if ((rc = ensure(c, ix->segs_dev, std::max<size_t>(1, ix->segs.size()) * sizeof(IndexSeg)))) return rc;
  if (!ix->segs.empty())
    HIP_TRY(c, hipMemcpyAsync(ix->segs_dev.p, ix->segs.data(), ix->segs.size() * sizeof(IndexSeg),
                              hipMemcpyHostToDevice, c->stream));
  return 0;
}

int index_reset_call(fqg_ctx* c) {
  IndexCall z;
  memset(&z, 0, sizeof(z));
  z.first_dup = z.first_wrong = z.first_missing = kNoRecord;
  *c->h_icall = z;
  HIP_TRY(c, hipMemcpyAsync(c->d_icall, c->h_icall, sizeof(IndexCall), hipMemcpyHostToDevice, c->stream));
  return 0;
}

int index_fetch_call(fqg_ctx* c) {
  HIP_TRY(c, hipMemcpyAsync(c->h_icall, c->d_icall, sizeof(IndexCall), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

unsigned index_grid(fqg_ctx* c, uint64_t n) {
  return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)c->cu_count * 16));
}

// the capture records of the last fqg_validate belong to this frame (FQG_VALIDATE_NAMES, streamed)
// (digests hold no name bytes and are made under one file state: they serve the insert into an index that keeps no name
// records, for that very state - everybody else reads the names through the line index)
bool names_usable(const fqg_ctx* c, const FrameView& fv, const fqg_file_state* st, bool bytes_needed) {
  static const bool off = getenv("FQGPU_NO_NAME_CAPTURE") != nullptr;  // (A/B: always go through the line index)
  if (off || !c->names_img || c->names_img != fv.img || c->names_nbytes != fv.nbytes) return false;
  if (c->names.digests) return !bytes_needed && st->readname_format == c->names.fmt && st->is_pe == c->names.is_pe;
  return true;
}
// the lists k_names_pass fills for k_names_rest
int env_int(const char* name, int dflt);
int names_prepare(fqg_ctx* c, const FrameView& fv) {
  int rc;
  (void)fv;
  const uint64_t n_slots = (uint64_t)c->names.cr.n_chunks << c->names.k_shift;
  if ((rc = ensure(c, c->name_redo, ((n_slots + 63) / 64 + 1) * 8))) return rc;
  if ((rc = ensure(c, c->name_redo_chunks, (((size_t)std::max<uint32_t>(c->names.cr.n_chunks, 1) + 15) & ~(size_t)15) + 16))) return rc;
  c->names.redo_bits = (unsigned long long*)c->name_redo.p;
  c->names.chunk_redo = (uint8_t*)c->name_redo_chunks.p;
  static const int abl = measure_int("FQGPU_NAMES_ABL");
  c->names.ablate = abl;
  return 0;
}
// (measurement knobs: FQGPU_NAMES_BLOCKS_PER_CU, FQGPU_NAMES_DYN_LDS - bytes of unused LDS per workgroup, which caps the
// workgroups a CU holds -, FQGPU_NAMES_NT=0 for plain instead of non-temporal loads of the capture records)
int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
unsigned names_grid(fqg_ctx* c) {
  static const int per_cu = std::max(1, env_int("FQGPU_NAMES_BLOCKS_PER_CU", 16));
  const uint64_t n_slots = (uint64_t)c->names.cr.n_chunks << c->names.k_shift;
  return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_slots + kBlock - 1) / kBlock, (uint64_t)c->cu_count * per_cu));
}
void launch_names_pass(fqg_ctx* c, bool match, const FrameView& fv, const IndexView& iv, const fqg_file_state* st,
                       uint64_t base, unsigned long long* found) {
  static const unsigned dyn_lds = (unsigned)std::max(0, env_int("FQGPU_NAMES_DYN_LDS", 0));
  static const bool nt = env_int("FQGPU_NAMES_NT", 1) != 0;
#define FQG_NAMES_PASS(M, N)                                                                                             \
  hipLaunchKernelGGL((k_names_pass<M, N>), dim3(names_grid(c)), dim3(kBlock), dyn_lds, c->stream, fv, c->names, iv,       \
                     st->readname_format, st->is_pe, base, found, c->d_icall)
  if (match) {
    if (nt) FQG_NAMES_PASS(true, true);
    else FQG_NAMES_PASS(true, false);
  } else if (c->names.digests) {
    hipLaunchKernelGGL((k_names_pass<false, true, true>), dim3(names_grid(c)), dim3(kBlock), dyn_lds, c->stream, fv, c->names, iv,
                       st->readname_format, st->is_pe, base, found, c->d_icall);
  } else {
    if (nt) FQG_NAMES_PASS(false, true);
    else FQG_NAMES_PASS(false, false);
  }
#undef FQG_NAMES_PASS
}

// The table built part by part in LDS from the name digests of the current frame (fqg_names_build_kernels.hip) instead of
// one CAS per name.  Returns 0 when it ran (the call's scalars hold its counts and findings), 1 when it does not apply
// or a bucket overflowed (nothing has touched the table: the caller takes k_names_pass), < 0 on errors.
constexpr unsigned long long kBuildSpillCap = 1ull << 20;
int names_build(fqg_ctx* c, fqg_index* ix, const FrameView& fv, const fqg_file_state* st, uint64_t record_base) {
  const int mode = env_int("FQGPU_NAMES_BUILD", -1);  // 0: never, 1: whenever the table allows it (tests), -1: by size
  if (mode == 0) return 1;
  const bool from_records = !c->names.digests;  // (an index that keeps name records: keys and names[] from the capture records)
  uint32_t k = 0;
  while ((1ull << k) < ix->capacity) ++k;
  if (k < kPartLogMin + 1) return 1;
  uint32_t part_log = std::min(std::max(k > 16 ? k - 16 : 0u, kPartLogMin), kPartLogMax);
  {
    const int forced = env_int("FQGPU_BUILD_PART_LOG", 0);  // (A/B)
    if (forced == 12 || forced == 13) part_log = std::max<uint32_t>(forced, k > 16 ? k - 16 : 0u);
    if (part_log > kPartLogMax || k < part_log + 1) return 1;
  }
  const uint32_t part_bits = k - part_log;
  const uint32_t bits0 = std::min<uint32_t>(part_bits, 8), bits1 = part_bits - bits0;
  if (bits1 > 8) return 1;
  // worth it when the frame brings a sizeable share of the table: the build writes (and, for an index that is not
  // empty, first reads) every part - 8 bytes per SLOT -, the CAS pass costs by the name
  const bool empty = ix->n_records_total == 0;
  if (mode < 0 && fv.n_records < ix->capacity / (empty ? 16 : 8)) return 1;
  const uint64_t n = fv.n_records;
  auto room = [](double mean) { return (unsigned long long)(mean + 8.0 * std::sqrt(mean + 1.0) + 64.0); };
  const unsigned long long cap0 = room((double)n / (double)(1u << bits0));
  const unsigned long long cap1 = bits1 ? room((double)n / (double)(1u << part_bits)) : 0;
  int rc;
  if ((rc = ensure(c, c->build_keys[0], (size_t)(cap0 << bits0) * sizeof(BuildKey)))) return rc;
  if (bits1 && (rc = ensure(c, c->build_keys[1], (size_t)(cap1 << part_bits) * sizeof(BuildKey)))) return rc;
  const size_t n_cursors = (size_t)(1u << bits0) + (bits1 ? (size_t)(1u << part_bits) : 0);
  if ((rc = ensure(c, c->build_cursor, n_cursors * 4))) return rc;
  if ((rc = ensure(c, c->build_spill, (size_t)kBuildSpillCap * sizeof(BuildKey)))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->build_cursor.p, 0, n_cursors * 4, c->stream));
  unsigned int* cur0 = (unsigned int*)c->build_cursor.p;
  unsigned int* cur1 = cur0 + (1u << bits0);
  const uint64_t mask = ix->capacity - 1;
  BuildLevel L0{(BuildKey*)c->build_keys[0].p, cur0, cap0, k - bits0, bits0};
  const uint64_t n_slots = (uint64_t)c->names.cr.n_chunks << c->names.k_shift;
  {
    ProfScope ps(c, "k_names_build_scatter0");
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_slots + kBuildTile - 1) / kBuildTile, (uint64_t)c->cu_count * 8));
    if (from_records)
      hipLaunchKernelGGL(k_build_scatter<2>, dim3(grid), dim3(kBuildThreads), 0, c->stream, fv, c->names, record_base, mask, L0,
                         (const BuildKey*)nullptr, (const unsigned int*)nullptr, 0ull, c->d_icall,
                         ix->keep_names ? (NameRec*)ix->names.p : (NameRec*)nullptr, st->readname_format, st->is_pe);
    else
      hipLaunchKernelGGL(k_build_scatter<0>, dim3(grid), dim3(kBuildThreads), 0, c->stream, fv, c->names, record_base, mask, L0,
                         (const BuildKey*)nullptr, (const unsigned int*)nullptr, 0ull, c->d_icall, (NameRec*)nullptr, 0, 0);
  }
  const BuildKey* part_keys = L0.out;
  const unsigned int* part_count = cur0;
  unsigned long long part_cap = cap0;
  if (bits1) {
    BuildLevel L1{(BuildKey*)c->build_keys[1].p, cur1, cap1, part_log, bits1};
    ProfScope ps(c, "k_names_build_scatter1");
    const unsigned tiles = (unsigned)((cap0 + kBuildTile - 1) / kBuildTile);
    hipLaunchKernelGGL(k_build_scatter<1>, dim3(tiles, 1u << bits0), dim3(kBuildThreads), 0, c->stream, fv, c->names, record_base, mask, L1,
                       (const BuildKey*)L0.out, (const unsigned int*)cur0, cap0, c->d_icall, (NameRec*)nullptr, 0, 0);
    part_keys = L1.out;
    part_count = cur1;
    part_cap = cap1;
  }
  {
    ProfScope ps(c, "k_names_build_parts");
    if (part_log == 12)
      hipLaunchKernelGGL(k_build_parts<12>, dim3(1u << part_bits), dim3(kBlock), 0, c->stream, fv, index_view(ix), record_base, part_keys,
                         part_count, part_cap, empty ? 0 : 1, (BuildKey*)c->build_spill.p, kBuildSpillCap, c->d_icall);
    else
      hipLaunchKernelGGL(k_build_parts<13>, dim3(1u << part_bits), dim3(kBlock), 0, c->stream, fv, index_view(ix), record_base, part_keys,
                         part_count, part_cap, empty ? 0 : 1, (BuildKey*)c->build_spill.p, kBuildSpillCap, c->d_icall);
  }
  {
    ProfScope ps(c, "k_names_build_spill");
    hipLaunchKernelGGL(k_build_spill, dim3(64), dim3(kBlock), 0, c->stream, fv, index_view(ix), record_base,
                       (const BuildKey*)c->build_spill.p, kBuildSpillCap, c->d_icall);
  }
  if ((rc = index_fetch_call(c))) return rc;
  // (gigabytes that only this call needs: a context that keeps them is what the next large allocation of the process
  // fails on - the bench's 200 M-pair leg did, with 4 GB free of 288)
  release(c->build_keys[0]);
  release(c->build_keys[1]);
  if (c->h_icall->build_overflow) return 1;
  return 0;
}

// rebuild the table at a larger capacity from the retained segments
int index_grow(fqg_index* ix, uint64_t need_names) {
  fqg_ctx* c = ix->ctx;
  uint64_t cap = ix->capacity ? ix->capacity : 1024;
  while (cap < 2 * need_names) cap <<= 1;
  if (cap == ix->capacity) return 0;
  int rc;
  if ((rc = index_alloc_table(ix, cap))) return rc;
  if ((rc = index_upload_segs(ix))) return rc;
  for (size_t s = 0; s < ix->segs.size(); ++s) {
    if (!ix->segs[s].n_records) continue;  // (a frame whose insert failed: kept for ownership only)
    if ((rc = index_reset_call(c))) return rc;
    FrameView fv = ix->frames[s]->fv;
    ProfScope ps(c, "k_index_insert(regrow)");
    IndexView v = index_view(ix);
    v.names = nullptr;  // (the name records of these frames are in place)
    hipLaunchKernelGGL(k_index_insert, dim3(index_grid(c, fv.n_records)), dim3(kBlock), 0, c->stream, fv, v,
                       ix->segs[s].record_base, c->d_icall);
  }
  return 0;
}

}  // namespace

int fqg_index_create(fqg_ctx* c, uint64_t expected_names, fqg_index** out) {
  if (!c || !out) return FQG_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  fqg_index* ix = new fqg_index();
  ix->ctx = c;
  uint64_t cap = 1024;
  while (cap < 2 * expected_names) cap <<= 1;
  int rc = index_alloc_table(ix, cap);
  if (rc) {
    fqg_index_destroy(ix);
    return rc;
  }
  *out = ix;
  return 0;
}

int fqg_index_expect_lookups(fqg_index* ix, int yes) {
  if (!ix) return FQG_ERR_ARG;
  if (ix->n_records_total) return fail(ix->ctx, FQG_ERR_STATE, "fqg_index_expect_lookups: the index is not empty");
  ix->keep_names = yes != 0;
  return 0;
}

void fqg_index_destroy(fqg_index* ix) {
  if (!ix) return;
  (void)hipStreamSynchronize(ix->ctx->stream);
  for (auto* f : ix->frames) fqg_frame_release(f);
  release(ix->slots);
  release(ix->names);
  release(ix->claims);
  release(ix->segs_dev);
  release(ix->found);
  release(ix->match);
  delete ix;
}

int fqg_index_insert_unique(fqg_ctx* c, fqg_index* ix, const fqg_file_state* st, fqg_index_result* out) {
  if (!c || !ix || !st || !out || ix->ctx != c) return FQG_ERR_ARG;
  if (!c->frame_valid) return fail(c, FQG_ERR_STATE, "no frame: call fqg_validate first");
  if (const int irc = index_now(c)) return irc;
  memset(out, 0, sizeof(*out));
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  fqg_frame* fr = nullptr;
  if ((rc = fqg_frame_retain(c, &fr))) return rc;
  if (ix->frames.empty()) {
    ix->fmt = st->readname_format;
    ix->is_pe = st->is_pe;
  }
  ix->flags |= fr->flags;
  IndexSeg sg;
  sg.img = fr->fv.img;
  sg.line_end = fr->fv.line_end;
  sg.nbytes = fr->fv.nbytes;
  sg.n_records = fr->fv.n_records;
  sg.record_base = ix->n_records_total;
  ix->frames.push_back(fr);
  ix->segs.push_back(sg);
  // on any failure below the frame stays registered with the index (freed by fqg_index_destroy), but its records do
  // not count: n_records_total / inserted only move once the insert has succeeded
  if (2 * (ix->inserted + sg.n_records) > ix->capacity) {
    // grow first (re-inserting the earlier segments), then insert this one
    ix->segs.pop_back();
    ix->frames.pop_back();
    rc = index_grow(ix, ix->inserted + sg.n_records);
    ix->frames.push_back(fr);
    ix->segs.push_back(sg);
    if (rc) {
      ix->segs.back().n_records = 0;
      return rc;
    }
  }
  auto drop = [&](int code) {
    ix->segs.back().n_records = 0;  // (kept for ownership only)
    return code;
  };
  if ((rc = index_reserve_records(ix, ix->n_records_total + sg.n_records))) return drop(rc);
  if ((rc = index_upload_segs(ix))) return drop(rc);
  if ((rc = index_reset_call(c))) return drop(rc);
  const bool captured = names_usable(c, fr->fv, st, ix->keep_names);
  if (sg.n_records && captured) {
    if ((rc = names_prepare(c, fr->fv))) return drop(rc);
    const int built = names_build(c, ix, fr->fv, st, sg.record_base);
    if (built < 0) return drop(built);
    if (built == 1) {
      if ((rc = index_reset_call(c))) return drop(rc);
      ProfScope ps(c, "k_names_insert");
      launch_names_pass(c, false, fr->fv, index_view(ix), st, sg.record_base, nullptr);
    }
    ProfScope ps(c, "k_names_rest");
    hipLaunchKernelGGL(k_names_rest<false>, dim3(c->cu_count * 4), dim3(kBlock), 0, c->stream, fr->fv, c->names, index_view(ix),
                       st->readname_format, st->is_pe, sg.record_base, (unsigned long long*)nullptr, c->d_icall);
  } else if (sg.n_records) {
    ProfScope ps(c, "k_index_insert");
    hipLaunchKernelGGL(k_index_insert, dim3(index_grid(c, sg.n_records)), dim3(kBlock), 0, c->stream, fr->fv,
                       index_view(ix), sg.record_base, c->d_icall);
  }
  if ((rc = index_fetch_call(c))) return drop(rc);
  if (hipGetLastError() != hipSuccess) return drop(fail(c, FQG_ERR_HIP, "k_index_insert"));
  if (c->h_icall->table_full) return drop(fail(c, FQG_ERR_STATE, "name index full"));
  if (sg.n_records && c->h_icall->seen != sg.n_records && !measured_wrong(captured && c->names.ablate)) return drop(fail(c, FQG_ERR_STATE, "name index: the pass did not meet every record once"));
  c->last_names_captured = c->h_icall->captured;
  ix->n_records_total += sg.n_records;
  ix->inserted += c->h_icall->inserted;
  ix->name_bytes += c->h_icall->name_bytes;
  const uint64_t w = c->h_icall->first_wrong;
  const uint64_t d = c->h_icall->first_dup == kNoRecord ? kNoRecord : c->h_icall->first_dup - sg.record_base;
  if (w != kNoRecord && w < d) {
    out->code = FQG_E_WRONG_HEADER;
    out->record = w;
  } else if (d != kNoRecord) {
    out->code = FQG_E_DUP_NAME;
    out->record = d;
  }
  if (w != kNoRecord || d != kNoRecord) ix->names_once = false;
  out->n_entries = ix->inserted;
  // sizeof(hashtable) + per entry sizeof(INDEX_ENTRY) + len + 1 + sizeof(hashnode)
  out->index_mem = 8 + ix->inserted * (16 + 1 + 24) + ix->name_bytes;
  return 0;
}

static int index_match_impl(fqg_ctx* c, fqg_index* ix, const fqg_file_state* st, uint64_t* match, fqg_index_result* out) {
  if (!c || !ix || !st || !out || ix->ctx != c) return FQG_ERR_ARG;
  if (!c->frame_valid) return fail(c, FQG_ERR_STATE, "no frame: call fqg_validate first");
  if (const int irc = index_now(c)) return irc;
  memset(out, 0, sizeof(*out));
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if (!ix->claims.p && (rc = index_reserve_records(ix, std::max<uint64_t>(ix->n_records_total, 1)))) return rc;
  if ((rc = index_upload_segs(ix))) return rc;
  if ((rc = index_reset_call(c))) return rc;
  const FrameView fv = c->frame;
  unsigned long long *d_slot = nullptr, *d_match = nullptr;
  if (match && fv.n_records) {
    if ((rc = ensure(c, ix->found, fv.n_records * 8))) return rc;
    if ((rc = ensure(c, ix->match, fv.n_records * 8))) return rc;
    d_slot = (unsigned long long*)ix->found.p;
    d_match = (unsigned long long*)ix->match.p;
  }
  if (fv.n_records) {
    if (names_usable(c, fv, st, true)) {
      if ((rc = names_prepare(c, fv))) return rc;
      {
        ProfScope ps(c, "k_names_match");
        launch_names_pass(c, true, fv, index_view(ix), st, ix->n_askers_total, d_slot);
      }
      ProfScope ps(c, "k_names_rest");
      hipLaunchKernelGGL(k_names_rest<true>, dim3(c->cu_count * 4), dim3(kBlock), 0, c->stream, fv, c->names, index_view(ix),
                         st->readname_format, st->is_pe, ix->n_askers_total, d_slot, c->d_icall);
    } else {
      ProfScope ps(c, "k_index_match_delete");
      hipLaunchKernelGGL(k_index_match_delete, dim3(index_grid(c, fv.n_records)), dim3(kBlock), 0, c->stream, fv,
                         index_view(ix), st->readname_format, st->is_pe, (c->frame_flags & kFlagNul) ? 1 : 0,
                         ix->n_askers_total, d_slot, c->d_icall);
    }
    if (match) {
      {
        ProfScope ps(c, "k_index_probe_resolve");
        hipLaunchKernelGGL(k_index_probe_resolve, dim3((unsigned)((fv.n_records + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                           c->stream, fv.n_records, (const unsigned long long*)d_slot, index_view(ix), ix->n_askers_total,
                           d_match);
      }
      HIP_TRY(c, hipMemcpyAsync(match, d_match, fv.n_records * 8, hipMemcpyDeviceToHost, c->stream));
    }
  }
  if ((rc = index_fetch_call(c))) return rc;
  HIP_TRY(c, hipGetLastError());
  if (fv.n_records && c->h_icall->seen != fv.n_records && !measured_wrong(c->names.ablate)) return fail(c, FQG_ERR_STATE, "name index: the pass did not meet every record once");
  c->last_names_captured = c->h_icall->captured;
  ix->matched += c->h_icall->matched;
  ix->n_askers_total += fv.n_records;
  const uint64_t w = c->h_icall->first_wrong, m = c->h_icall->first_missing;
  if (w != kNoRecord && w < m) {
    out->code = FQG_E_WRONG_HEADER;
    out->record = w;
  } else if (m != kNoRecord) {
    out->code = FQG_E_UNPAIRED;
    out->record = m;
  }
  out->n_entries = ix->inserted - ix->matched;
  out->index_mem = 8 + ix->inserted * (16 + 1 + 24) + ix->name_bytes;
  return 0;
}

int fqg_index_match_delete(fqg_ctx* c, fqg_index* ix, const fqg_file_state* st, fqg_index_result* out) {
  return index_match_impl(c, ix, st, nullptr, out);
}

int fqg_index_probe_delete(fqg_ctx* c, fqg_index* ix, const fqg_file_state* st, uint64_t* match, fqg_index_result* out) {
  if (!match && c && c->frame_valid && c->frame.n_records) return FQG_ERR_ARG;
  return index_match_impl(c, ix, st, match, out);
}

int fqg_index_alive(fqg_ctx* c, fqg_index* ix, uint8_t* alive, uint64_t cap) {
  if (!c || !ix || ix->ctx != c || (!alive && cap)) return FQG_ERR_ARG;
  if (cap < ix->n_records_total) return fail(c, FQG_ERR_ARG, "fqg_index_alive: buffer too small");
  if (!ix->n_records_total) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = index_upload_segs(ix))) return rc;
  uint8_t* d = nullptr;
  if (hipMalloc((void**)&d, ix->n_records_total) != hipSuccess) return fail(c, FQG_ERR_NOMEM, "fqg_index_alive: device allocation failed");
  (void)hipMemsetAsync(d, 0, ix->n_records_total, c->stream);
  IndexView v = index_view(ix);
  hipLaunchKernelGGL(k_index_alive, dim3((unsigned)((ix->capacity + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, v,
                     ix->n_records_total, d);
  hipError_t e = hipMemcpyAsync(alive, d, ix->n_records_total, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(c, FQG_ERR_HIP, "fqg_index_alive", e);
  return 0;
}

int fqg_names_compare(fqg_ctx* c, const fqg_frame* a, const fqg_file_state* sa, const fqg_frame* b,
                      const fqg_file_state* sb, fqg_index_result* out) {
  if (!c || !a || !sa || !out || (b && !sb)) return FQG_ERR_ARG;
  memset(out, 0, sizeof(*out));
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = index_reset_call(c))) return rc;
  const int interleaved = b ? 0 : 1;
  const uint64_t n_pairs = interleaved ? a->fv.n_records / 2 : std::min(a->fv.n_records, b->fv.n_records);
  if (n_pairs) {
    ProfScope ps(c, "k_names_compare");
    const uint32_t fl = a->flags | (b ? b->flags : 0u);
    hipLaunchKernelGGL(k_names_compare, dim3(index_grid(c, n_pairs)), dim3(kBlock), 0, c->stream, a->fv,
                       sa->readname_format, sa->is_pe, b ? b->fv : a->fv, b ? sb->readname_format : sa->readname_format,
                       b ? sb->is_pe : sa->is_pe, interleaved, (fl & kFlagNul) ? 1 : 0, n_pairs, c->d_icall);
  }
  if ((rc = index_fetch_call(c))) return rc;
  HIP_TRY(c, hipGetLastError());
  const uint64_t w = c->h_icall->first_wrong, m = c->h_icall->first_missing;
  if (w != kNoRecord && w <= m) {
    out->code = FQG_E_WRONG_HEADER;
    out->record = interleaved ? 2 * w : w;
  } else if (m != kNoRecord) {
    out->code = interleaved ? FQG_E_UNPAIRED : FQG_E_NAME_MISMATCH;
    out->record = interleaved ? 2 * m : m;
  }
  return 0;
}

// ---- barcode extraction ----------------------------------------------------------------------
// Tile geometry for one call: T iterations whose records and output text fit `budget` bytes of LDS
// per wavefront, from the mean record sizes (a tile that does not fit anyway takes the direct path).
// FQGPU_BC_LDS overrides the budget (bytes, 4096..65536).
static BcTile bc_tile_for(const BcParams& P, unsigned default_budget = 20480u) {
  static const unsigned env_budget = [] {
    const char* e = getenv("FQGPU_BC_LDS");
    const long v = e ? atol(e) : 0;
    return (unsigned)(v >= 4096 && v <= 65536 ? v : 0);
  }();
  const unsigned budget = env_budget ? env_budget : default_budget;
  double in = 0, out_sam = 0, out_fq = 0;
  int files = 0;
  for (int x = 1; x < kBcFiles; ++x) {
    if (!P.f[x].present) continue;
    const double rec = P.f[x].fv.n_records ? (double)P.f[x].fv.nbytes / (double)P.f[x].fv.n_records : 0.0;
    in += rec * P.f[x].step;
    ++files;
    if (x <= 2) {
      out_sam += 1.35 * rec + 130;
      if (P.emit[x]) out_fq = std::max(out_fq, rec + 100);
    }
  }
  const double out = P.out_sam ? out_sam : out_fq;
  const unsigned lanes_per_iter = P.out_sam && P.f[2].present ? 2 : 1;
  const double fixed = 48.0 * files + 96.0;
  double t = ((double)budget - fixed) / (1.06 * (in + out) + 1.0);
  unsigned T = (unsigned)std::max(1.0, std::min(t, 64.0 / lanes_per_iter));
  BcTile tc;
  tc.T = T;
  tc.in_cap = ((unsigned)(1.06 * in * T + 48.0 * files + 48.0) + 15u) & ~15u;
  if (tc.in_cap + 1024u > budget) tc.in_cap = (budget / 2) & ~15u;
  tc.out_cap = (budget - tc.in_cap) & ~15u;
  // the plan kernel: several tiles at once while one lane per iteration allows it.  It stages only the files a
  // barcode is cut from, and only when a minimum quality makes it read their bytes (bc_staged<PLAN>)
  tc.plan_m = std::max(1u, std::min(64u / tc.T, 4u));
  double plan_in = 0;
  int plan_files = 0;
  for (int x = 1; x < kBcFiles; ++x)
    if (P.f[x].present && P.min_qual > 0 && (P.umi_read == x || P.cell_read == x || P.sample_read == x)) {
      plan_in += (P.f[x].fv.n_records ? (double)P.f[x].fv.nbytes / (double)P.f[x].fv.n_records : 0.0) * P.f[x].step;
      ++plan_files;
    }
  tc.plan_cap = ((unsigned)(1.06 * plan_in * T * tc.plan_m + 48.0 * plan_files + 64.0) + 15u) & ~15u;
  while (tc.plan_m > 1 && tc.plan_cap > 32768u) {  // (long reads in a barcode file)
    tc.plan_cap = (tc.plan_cap / tc.plan_m * (tc.plan_m - 1) + 15u) & ~15u;
    --tc.plan_m;
  }
  tc.plan_cap = std::min(tc.plan_cap, 65536u - 256u);
  return tc;
}

// the kernels' view of one batch (shared by the transform and the census of what it kept)
static int bc_make_params(fqg_ctx* c, const fqg_frame* const frames[6], const fqg_file_state states[6],
                          const uint64_t first_record[6], const fqg_barcode_params* bp, uint64_t n_iter,
                          uint64_t first_read_number, BcParams& P, bool& inter) {
  memset(&P, 0, sizeof(P));
  inter = bp->interleaved[0] != 0 || bp->interleaved[1] != 0;
  for (int x = 1; x < kBcFiles; ++x) {
    if (!bp->present[x]) continue;
    if (!frames[x]) return fail(c, FQG_ERR_ARG, "fqg_barcodes_transform: missing frame");
    BcFile& f = P.f[x];
    f.fv = frames[x]->fv;
    // NUL bytes (and the "\0\n" behind a piece of a line cut at the gzgets limits): lines are C strings (bc_clip_nul)
    f.has_nul = (frames[x]->flags & kFlagNul) ? 1 : 0;
    P.has_nul |= f.has_nul;
    f.first = first_record[x];
    f.present = 1;
    f.fmt = states[x].readname_format;
    f.step = 1;
    f.add = 0;
    if (inter && (x == bp->interleaved[0] || x == bp->interleaved[1])) {
      f.step = 2;
      f.add = x == bp->interleaved[1] ? 1 : 0;
    }
    // every record the batch touches must exist
    if (n_iter && f.first + (n_iter - 1) * f.step + f.add >= f.fv.n_records)
      return fail(c, FQG_ERR_ARG, "fqg_barcodes_transform: iterations beyond a frame");
    P.n_inputs++;
  }
  if (!P.f[1].present) return fail(c, FQG_ERR_ARG, "fqg_barcodes_transform: read1 is required");
  for (int x = 1; x <= 2; ++x)
    if (((bp->out_sam && x == 1) || (!bp->out_sam && bp->emit[x])) && !P.f[x].present)
      return fail(c, FQG_ERR_ARG, "fqg_barcodes_transform: output requested for an absent input");
  const int64_t sizes[3] = {bp->umi_size, bp->cell_size, bp->sample_size};
  for (int i = 0; i < 3; ++i)
    if (sizes[i] < 0 || sizes[i] >= 50) return fail(c, FQG_ERR_ARG, "barcode sizes must be 0..49 (MAX_BARCODE_LENGTH)");
  P.umi_read = bp->umi_read;
  P.cell_read = bp->cell_read;
  P.sample_read = bp->sample_read;
  P.phred = bp->phred_encoding;
  P.min_qual = bp->min_qual;
  P.out_sam = bp->out_sam;
  P.tenx = bp->tenx;
  {
    static const int abl = measure_int("FQGPU_BC_ABL");
    P.ablate = abl;
  }
  P.umi_off = bp->umi_offset;
  P.umi_size = bp->umi_size;
  P.cell_off = bp->cell_offset;
  P.cell_size = bp->cell_size;
  P.sample_off = bp->sample_offset;
  P.sample_size = bp->sample_size;
  for (int x = 0; x < 3; ++x) {
    P.emit[x] = bp->emit[x];
    P.read_off[x] = bp->read_offset[x];
    P.read_size[x] = bp->read_size[x];
  }
  P.first_read_number = first_read_number;
  return 0;
}

int fqg_barcodes_transform(fqg_ctx* c, const fqg_frame* const frames[6], const fqg_file_state states[6],
                           const uint64_t first_record[6], const fqg_barcode_params* bp, uint64_t n_iter,
                           uint64_t first_read_number, fqg_barcode_result* out) {
  if (!c || !frames || !states || !first_record || !bp || !out) return FQG_ERR_ARG;
  if (c->out_pending) {  // (a copy of the previous output is still on its way: this call writes the same buffers)
    const int rcw = fqg_barcodes_output_wait(c);
    if (rcw) return rcw;
  }
  memset(out, 0, sizeof(*out));
  c->bc_out_bytes[0] = c->bc_out_bytes[1] = c->bc_out_bytes[2] = 0;
  c->bc_status_valid = 0;
  HIP_TRY(c, hipSetDevice(c->device));
  BcParams P;
  bool inter;
  int rc;
  if ((rc = bc_make_params(c, frames, states, first_record, bp, n_iter, first_read_number, P, inter))) return rc;
  if (!n_iter) return 0;

  if ((rc = ensure(c, c->bc_status, n_iter))) return rc;
  const uint64_t nb = (n_iter + kScan64Span - 1) / kScan64Span;
  for (int i = 0; i < 3; ++i) {
    if ((rc = ensure(c, c->bc_len[i], n_iter * 4))) return rc;
    if ((rc = ensure(c, c->bc_off[i], n_iter * 8))) return rc;
    if ((rc = ensure(c, c->bc_sum[i], nb * 8))) return rc;
  }
  BcCall z;
  memset(&z, 0, sizeof(z));
  z.first_finding = z.first_discard = ~0ull;
  *c->h_bcall = z;
  HIP_TRY(c, hipMemcpyAsync(c->d_bcall, c->h_bcall, sizeof(BcCall), hipMemcpyHostToDevice, c->stream));
  const BcTile tc = bc_tile_for(P);
  int file_mask = 0;  // the usual sets of inputs have kernels of their own (bc_has)
  for (int x = 1; x < kBcFiles; ++x)
    if (P.f[x].present) file_mask |= 1 << x;
  const uint64_t n_tiles = (n_iter + tc.T - 1) / tc.T;
  if ((rc = ensure(c, c->bc_tile_big, n_tiles))) return rc;
  // persistent grids: exactly the wavefronts that are resident at once (tiles are dealt round-robin,
  // so a wavefront that starts late would do its whole share after the others have finished)
  auto resident = [&](const void* kernel, unsigned lds) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kWave, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    return (unsigned)per_cu * (unsigned)c->cu_count;
  };
  {
    ProfScope ps(c, "k_bc_plan");
#define FQG_PLAN_TILE(MASK)                                                                                       \
  do {                                                                                                            \
    const unsigned plan_lds = tc.plan_cap;                                                                        \
    const uint64_t plan_tiles = (n_tiles + tc.plan_m - 1) / tc.plan_m;                                            \
    const unsigned grid = (unsigned)std::min<uint64_t>(plan_tiles, resident((const void*)k_bc_plan_tile<MASK>, plan_lds)); \
    hipLaunchKernelGGL(k_bc_plan_tile<MASK>, dim3(grid), dim3(kWave), plan_lds, c->stream, P, tc, n_iter,          \
                       (uint8_t*)c->bc_status.p, (uint32_t*)c->bc_len[0].p, (uint32_t*)c->bc_len[1].p,            \
                       (uint32_t*)c->bc_len[2].p, (uint8_t*)c->bc_tile_big.p, c->d_bcall);                         \
  } while (0)
    switch (file_mask) {
      case 0x02: FQG_PLAN_TILE(0x02); break;
      case 0x06: FQG_PLAN_TILE(0x06); break;
      case 0x0A: FQG_PLAN_TILE(0x0A); break;
      case 0x0E: FQG_PLAN_TILE(0x0E); break;
      default: FQG_PLAN_TILE(0); break;
    }
#undef FQG_PLAN_TILE
  }
  HIP_TRY(c, hipMemcpyAsync(c->h_bcall, c->d_bcall, sizeof(BcCall), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  uint64_t n_done = n_iter;
  const uint64_t n_big = c->h_bcall->big;
  // interleaved input re-synchronises behind the first discard: nothing beyond it is reached in this batch (a name
  // finding - the emit kernels look for those - can only lie in what is emitted)
  if (inter && c->h_bcall->first_discard < n_done) n_done = c->h_bcall->first_discard + 1;
  out->n_done = n_done;
  unsigned long long* d_tot = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->d_bcall) + sizeof(BcCall));
  unsigned long long* h_tot = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->h_bcall) + sizeof(BcCall));
  const uint64_t nb_done = (n_done + kScan64Span - 1) / kScan64Span;
  {
    ProfScope ps(c, "k_bc_scan");
    if (n_done != n_iter) {  // (the plan counted the discards of all n_iter iterations)
      HIP_TRY(c, hipMemsetAsync(&c->d_bcall->discarded, 0, 2 * sizeof(unsigned long long), c->stream));
      hipLaunchKernelGGL(k_bc_count, dim3((unsigned)std::min<uint64_t>((n_done + kBlock - 1) / kBlock, 2048)), dim3(kBlock), 0,
                         c->stream, (const uint8_t*)c->bc_status.p, n_done, c->d_bcall);
    }
    HIP_TRY(c, hipMemsetAsync(d_tot, 0, 3 * sizeof(unsigned long long), c->stream));
    for (int i = 0; i < 3; ++i) {
      if (P.out_sam ? i != 0 : !P.emit[i]) continue;  // outputs that are not produced have no lengths
      hipLaunchKernelGGL(k_scan64_a, dim3((unsigned)nb_done), dim3(kBlock), 0, c->stream, (const uint32_t*)c->bc_len[i].p,
                         n_done, (unsigned long long*)c->bc_off[i].p, (unsigned long long*)c->bc_sum[i].p);
      hipLaunchKernelGGL(k_scan64_b, dim3(1), dim3(kBlock), 0, c->stream, (unsigned long long*)c->bc_sum[i].p, nb_done,
                         d_tot + i);
    }
  }
  HIP_TRY(c, hipMemcpyAsync(c->h_bcall, c->d_bcall, sizeof(BcCall) + 64, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  out->n_discarded = c->h_bcall->discarded;
  out->n_short = c->h_bcall->short_warnings;
  for (int i = 0; i < 3; ++i) {
    out->out_bytes[i] = h_tot[i];
    c->bc_out_bytes[i] = h_tot[i];
    if ((rc = ensure(c, c->bc_out[i], std::max<uint64_t>(h_tot[i], 16)))) return rc;
  }
  {
    ProfScope ps(c, "k_bc_emit");
    EmitOut eo[3];
    for (int i = 0; i < 3; ++i)
      eo[i] = EmitOut{(const uint32_t*)c->bc_len[i].p, (const unsigned long long*)c->bc_off[i].p,
                      (const unsigned long long*)c->bc_sum[i].p, (uint8_t*)c->bc_out[i].p};
    const uint64_t n_tiles_done = (n_done + tc.T - 1) / tc.T;
    const unsigned lds = tc.in_cap + tc.out_cap;
    unsigned grid_t = 1;
#define FQG_EMIT_TILE(SAM, MASK)                                                                                  \
  do {                                                                                                            \
    grid_t = (unsigned)std::min<uint64_t>(n_tiles_done, resident((const void*)k_bc_emit_tile<SAM, MASK>, lds));    \
    hipLaunchKernelGGL((k_bc_emit_tile<SAM, MASK>), dim3(grid_t), dim3(kWave), lds, c->stream, P, tc, n_done,      \
                       (const uint8_t*)c->bc_status.p, (const uint8_t*)c->bc_tile_big.p, eo[0], eo[1], eo[2],      \
                       c->d_bcall);                                                                                \
  } while (0)
#define FQG_EMIT_TILE_MASKS(SAM)                  \
  switch (file_mask) {                            \
    case 0x02: FQG_EMIT_TILE(SAM, 0x02); break;   \
    case 0x06: FQG_EMIT_TILE(SAM, 0x06); break;   \
    case 0x0A: FQG_EMIT_TILE(SAM, 0x0A); break;   \
    case 0x0E: FQG_EMIT_TILE(SAM, 0x0E); break;   \
    default: FQG_EMIT_TILE(SAM, 0); break;        \
  }
    if (P.out_sam) FQG_EMIT_TILE_MASKS(true)
    else FQG_EMIT_TILE_MASKS(false)
#undef FQG_EMIT_TILE_MASKS
#undef FQG_EMIT_TILE
    if (getenv("FQGPU_BC_DEBUG"))
      fprintf(stderr, "fqgpu barcodes: T=%u in_cap=%u out_cap=%u emit grid=%u (%u/CU) big tiles=%llu\n", tc.T, tc.in_cap,
              tc.out_cap, grid_t, grid_t / (unsigned)c->cu_count, (unsigned long long)n_big);
    if (n_big) {
      const unsigned grid_e =
          (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_done + 3) / 4, (uint64_t)c->cu_count * 8));
      hipLaunchKernelGGL(k_bc_emit_direct, dim3(grid_e), dim3(kBlock), 0, c->stream, P, tc, n_done,
                         (const uint8_t*)c->bc_status.p, (const uint8_t*)c->bc_tile_big.p, eo[0], eo[1], eo[2], c->d_bcall);
    }
  }
  HIP_TRY(c, hipMemcpyAsync(c->h_bcall, c->d_bcall, sizeof(BcCall), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  if (c->h_bcall->first_finding != ~0ull) {
    // Names that do not agree (or a header without '@') at iteration k: the reference stops there, before it looks at
    // the iteration's barcodes.  What the batch yields is what lies in front of k: the text up to k's place in every
    // output, the discards among the first k iterations.
    const uint64_t k = c->h_bcall->first_finding >> 8;
    out->iteration = k;
    out->code = (int32_t)((c->h_bcall->first_finding & 0xFF) >> 3);
    out->file = (int32_t)(c->h_bcall->first_finding & 7);
    out->n_done = k;
    out->n_discarded = out->n_short = 0;
    for (int i = 0; i < 3; ++i) {
      if (P.out_sam ? i != 0 : !P.emit[i]) continue;
      unsigned long long off = 0, sum = 0;
      if (k) {  // (k < n_done: both entries exist)
        HIP_TRY(c, hipMemcpy(&off, (const unsigned long long*)c->bc_off[i].p + k, 8, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(&sum, (const unsigned long long*)c->bc_sum[i].p + k / kScan64Span, 8, hipMemcpyDeviceToHost));
      }
      out->out_bytes[i] = off + sum;
      c->bc_out_bytes[i] = off + sum;
    }
    if (k) {
      HIP_TRY(c, hipMemsetAsync(&c->d_bcall->discarded, 0, 2 * sizeof(unsigned long long), c->stream));
      hipLaunchKernelGGL(k_bc_count, dim3((unsigned)std::min<uint64_t>((k + kBlock - 1) / kBlock, 2048)), dim3(kBlock), 0,
                         c->stream, (const uint8_t*)c->bc_status.p, k, c->d_bcall);
      HIP_TRY(c, hipMemcpyAsync(c->h_bcall, c->d_bcall, sizeof(BcCall), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      out->n_discarded = c->h_bcall->discarded;
      out->n_short = c->h_bcall->short_warnings;
    }
  }
  c->bc_status_valid = out->n_done;  // (fqg_barcodes_census: the status bytes of the iterations this batch yielded)
  return 0;
}

// ---- census of what the transform kept (fqg_census_kernels.hip) -------------------------------------
struct fqg_census {
  fqg_ctx* ctx = nullptr;
  DevBuf cells, umis;     // the pairs, in arrival order until fqg_census_finish sorts them
  DevBuf cells2, umis2, tmp, flag, local, sums, lines;
  unsigned long long* d_count = nullptr;
  uint64_t n_pairs = 0, n_cells = 0;
  unsigned umi_bits = 1, cell_bits = 1;  // bits the packed values of the batches so far can need (census_bits)
  bool finished = false;
};

// char2uint_64 of at most `size` characters is below 10^size: the sorts of fqg_census_finish need not look at more bits
// (a UMI of 10 bases: 34 of 64, a cell barcode of 16: 54).  A size that is not known (< 0), or beyond 19 digits: all 64.
static unsigned census_bits(long size) {
  if (size <= 0) return size == 0 ? 1u : 64u;
  if (size > 19) return 64u;
  unsigned long long lim = 1;
  for (long i = 0; i < size; ++i) lim *= 10ull;
  unsigned bits = 1;
  while (bits < 64 && (1ull << bits) < lim) ++bits;
  return bits;
}

int fqg_census_create(fqg_ctx* c, fqg_census** out) {
  if (!c || !out) return FQG_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  std::unique_ptr<fqg_census> z(new fqg_census());
  z->ctx = c;
  if (hipMalloc((void**)&z->d_count, 8) != hipSuccess) return fail(c, FQG_ERR_NOMEM, "fqg_census_create");
  HIP_TRY(c, hipMemsetAsync(z->d_count, 0, 8, c->stream));
  *out = z.release();
  return 0;
}

void fqg_census_destroy(fqg_census* z) {
  if (!z) return;
  (void)hipStreamSynchronize(z->ctx->stream);
  for (DevBuf* b : {&z->cells, &z->umis, &z->cells2, &z->umis2, &z->tmp, &z->flag, &z->local, &z->sums, &z->lines}) release(*b);
  if (z->d_count) (void)hipFree(z->d_count);
  delete z;
}

// room for `need` pairs in a pair of arrays that keep their contents
static int census_reserve(fqg_ctx* c, fqg_census* z, uint64_t need) {
  if (need * 8 <= z->cells.cap) return 0;
  const uint64_t cap = std::max<uint64_t>(need + need / 2, 1u << 16);
  DevBuf a, b;
  if (hipMalloc(&a.p, cap * 8) != hipSuccess) return fail(c, FQG_ERR_NOMEM, "fqg_barcodes_census");
  if (hipMalloc(&b.p, cap * 8) != hipSuccess) {
    (void)hipFree(a.p);
    return fail(c, FQG_ERR_NOMEM, "fqg_barcodes_census");
  }
  a.cap = b.cap = cap * 8;
  if (z->n_pairs) {
    (void)hipMemcpyAsync(a.p, z->cells.p, z->n_pairs * 8, hipMemcpyDeviceToDevice, c->stream);
    (void)hipMemcpyAsync(b.p, z->umis.p, z->n_pairs * 8, hipMemcpyDeviceToDevice, c->stream);
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  release(z->cells);
  release(z->umis);
  z->cells = a;
  z->umis = b;
  return 0;
}

int fqg_barcodes_census(fqg_ctx* c, fqg_census* z, const fqg_frame* const frames[6], const fqg_file_state states[6],
                        const uint64_t first_record[6], const fqg_barcode_params* bp, uint64_t n_done, uint64_t* n_added) {
  if (!c || !z || z->ctx != c || !frames || !states || !first_record || !bp) return FQG_ERR_ARG;
  if (z->finished) return fail(c, FQG_ERR_STATE, "fqg_barcodes_census: the census is finished");
  if (n_done > c->bc_status_valid)
    return fail(c, FQG_ERR_STATE, "fqg_barcodes_census: call it behind the fqg_barcodes_transform of the same batch, with that call's n_done");
  if (n_added) *n_added = 0;
  if (!n_done) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  BcParams P;
  bool inter;
  int rc;
  if ((rc = bc_make_params(c, frames, states, first_record, bp, n_done, 0, P, inter))) return rc;
  if ((rc = census_reserve(c, z, z->n_pairs + n_done))) return rc;
  z->umi_bits = std::max(z->umi_bits, census_bits((long)P.umi_size));
  z->cell_bits = std::max(z->cell_bits, P.cell_read > 0 ? census_bits((long)P.cell_size) : 1u);
  HIP_TRY(c, hipMemsetAsync(z->d_count, 0, 8, c->stream));
  {
    ProfScope ps(c, "k_bc_census");
    const uint64_t tile = (uint64_t)kBlock * kCensusPer;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_done + tile - 1) / tile, (uint64_t)c->cu_count * 16));
    hipLaunchKernelGGL(k_bc_census, dim3(grid), dim3(kBlock), 0, c->stream, P, n_done, (const uint8_t*)c->bc_status.p,
                       (unsigned long long*)z->cells.p, (unsigned long long*)z->umis.p, (unsigned long long)z->n_pairs,
                       (unsigned long long)(z->cells.cap / 8), z->d_count);
  }
  unsigned long long added = 0;
  HIP_TRY(c, hipMemcpyAsync(&added, z->d_count, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  z->n_pairs += added;
  if (n_added) *n_added = added;
  return 0;
}

int fqg_census_finish(fqg_ctx* c, fqg_census* z, uint64_t* n_pairs, uint64_t* n_cells) {
  if (!c || !z || z->ctx != c) return FQG_ERR_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t n = z->n_pairs;
  if (!z->finished && n) {
    int rc;
    if ((rc = ensure(c, z->cells2, n * 8)) || (rc = ensure(c, z->umis2, n * 8))) return rc;
    if (n >= (1ull << 31)) return fail(c, FQG_ERR_ARG, "fqg_census_finish: more than 2^31 pairs");
    // by UMI, then (stable) by cell: sorted by (cell, UMI)
    unsigned long long *ce = (unsigned long long*)z->cells.p, *um = (unsigned long long*)z->umis.p;
    unsigned long long *ce2 = (unsigned long long*)z->cells2.p, *um2 = (unsigned long long*)z->umis2.p;
    size_t tmp_bytes = 0;
    const unsigned ub = z->umi_bits, cb = z->cell_bits;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, um, um2, ce, ce2, n, 0, 64, c->stream) != hipSuccess)
      return fail(c, FQG_ERR_HIP, "fqg_census_finish: sort");
    if ((rc = ensure(c, z->tmp, tmp_bytes + 16))) return rc;
    ProfScope ps(c, "k_census_finish");
    if (rocprim::radix_sort_pairs(z->tmp.p, tmp_bytes, um, um2, ce, ce2, n, 0, ub, c->stream) != hipSuccess ||
        rocprim::radix_sort_pairs(z->tmp.p, tmp_bytes, ce2, ce, um2, um, n, 0, cb, c->stream) != hipSuccess)
      return fail(c, FQG_ERR_HIP, "fqg_census_finish: sort");
    const uint64_t nb = (n + kScan64Span - 1) / kScan64Span;
    if ((rc = ensure(c, z->flag, n * 4)) || (rc = ensure(c, z->local, n * 8)) || (rc = ensure(c, z->sums, nb * 8 + 8))) return rc;
    unsigned long long* d_total = (unsigned long long*)z->sums.p + nb;
    HIP_TRY(c, hipMemsetAsync(d_total, 0, 8, c->stream));
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_census_flags, dim3(grid), dim3(kBlock), 0, c->stream, (const unsigned long long*)ce, n, (uint32_t*)z->flag.p);
    hipLaunchKernelGGL(k_scan64_a, dim3((unsigned)nb), dim3(kBlock), 0, c->stream, (const uint32_t*)z->flag.p, n,
                       (unsigned long long*)z->local.p, (unsigned long long*)z->sums.p);
    hipLaunchKernelGGL(k_scan64_b, dim3(1), dim3(kBlock), 0, c->stream, (unsigned long long*)z->sums.p, nb, d_total);
    unsigned long long cells = 0;
    HIP_TRY(c, hipMemcpyAsync(&cells, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    z->n_cells = cells;
    if ((rc = ensure(c, z->lines, cells * sizeof(CensusCell)))) return rc;
    HIP_TRY(c, hipMemsetAsync(z->lines.p, 0, cells * sizeof(CensusCell), c->stream));
    hipLaunchKernelGGL(k_census_count, dim3(grid), dim3(kBlock), 0, c->stream, (const unsigned long long*)ce,
                       (const unsigned long long*)um, n, (const uint32_t*)z->flag.p, (const unsigned long long*)z->local.p,
                       (const unsigned long long*)z->sums.p, (CensusCell*)z->lines.p);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
  }
  z->finished = true;
  if (n_pairs) *n_pairs = z->n_pairs;
  if (n_cells) *n_cells = z->n_cells;
  return 0;
}

int fqg_census_cells(fqg_ctx* c, fqg_census* z, fqg_census_cell* out, uint64_t cap) {
  if (!c || !z || z->ctx != c || (!out && cap)) return FQG_ERR_ARG;
  if (!z->finished) return fail(c, FQG_ERR_STATE, "fqg_census_cells: fqg_census_finish first");
  static_assert(sizeof(fqg_census_cell) == sizeof(CensusCell), "the C-ABI line is the device line");
  const uint64_t n = std::min<uint64_t>(cap, z->n_cells);
  if (n) HIP_TRY(c, hipMemcpy(out, z->lines.p, n * sizeof(CensusCell), hipMemcpyDeviceToHost));
  return 0;
}

int fqg_census_pairs(fqg_ctx* c, fqg_census* z, uint64_t* cells, uint64_t* umis, uint64_t cap) {
  if (!c || !z || z->ctx != c || ((!cells || !umis) && cap)) return FQG_ERR_ARG;
  const uint64_t n = std::min<uint64_t>(cap, z->n_pairs);
  if (n) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(cells, z->cells.p, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(umis, z->umis.p, n * 8, hipMemcpyDeviceToHost));
  }
  return 0;
}

const void* fqg_census_device_pairs(const fqg_census* z, int which) {
  return z ? (which ? z->umis.p : z->cells.p) : nullptr;
}

// ---- whitelist membership ---------------------------------------------------------------------
struct fqg_whitelist {
  fqg_ctx* ctx = nullptr;
  DevBuf set;
  uint64_t capacity = 0;
};

int fqg_whitelist_create(fqg_ctx* c, const uint64_t* packed, uint64_t n, fqg_whitelist** out) {
  if (!c || !out || (!packed && n)) return FQG_ERR_ARG;
  *out = nullptr;
  HIP_TRY(c, hipSetDevice(c->device));
  uint64_t cap = 1024;
  while (cap < 4 * n) cap <<= 1;
  std::vector<WlSlot> host(cap, WlSlot{0, 0});
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t at = wl_hash(packed[i]) & (cap - 1);
    while (host[at].used && host[at].key != packed[i]) at = (at + 1) & (cap - 1);
    host[at] = WlSlot{packed[i], 1};
  }
  std::unique_ptr<fqg_whitelist> w(new fqg_whitelist());
  w->ctx = c;
  w->capacity = cap;
  int rc;
  if ((rc = ensure(c, w->set, cap * sizeof(WlSlot)))) return rc;
  HIP_TRY(c, hipMemcpy(w->set.p, host.data(), cap * sizeof(WlSlot), hipMemcpyHostToDevice));
  *out = w.release();
  return 0;
}

void fqg_whitelist_destroy(fqg_whitelist* w) {
  if (!w) return;
  (void)hipStreamSynchronize(w->ctx->stream);
  release(w->set);
  delete w;
}

int fqg_barcodes_whitelist(fqg_ctx* c, const fqg_frame* frame, uint64_t first, uint64_t step, uint64_t n, int64_t offset,
                           int64_t size, const fqg_whitelist* wl, uint8_t* valid, fqg_whitelist_result* out) {
  if (!c || !frame || !wl || !out || wl->ctx != c || frame->ctx != c) return FQG_ERR_ARG;
  memset(out, 0, sizeof(*out));
  if (offset < 0 || size < 1 || size > kWlMaxSize) return fail(c, FQG_ERR_ARG, "fqg_barcodes_whitelist: offset >= 0 and 1 <= size <= 56");
  if (frame->flags & kFlagNul) return fail(c, FQG_ERR_ARG, "fqg_barcodes_whitelist: input holds NUL bytes");
  if (step < 1 || (n && first + (n - 1) * step >= frame->fv.n_records)) return fail(c, FQG_ERR_ARG, "fqg_barcodes_whitelist: records beyond the frame");
  out->n_records = n;
  if (!n) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  uint8_t* d_valid = nullptr;
  if (valid) {
    if ((rc = ensure(c, c->bc_status, n))) return rc;
    d_valid = (uint8_t*)c->bc_status.p;
  }
  WlCall* d_call = reinterpret_cast<WlCall*>(c->d_bcall);
  HIP_TRY(c, hipMemsetAsync(d_call, 0, sizeof(WlCall), c->stream));
  {
    ProfScope ps(c, "k_bc_whitelist");
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)c->cu_count * 32));
#define FQG_WL(W)                                                                                                       \
  hipLaunchKernelGGL(k_bc_whitelist<W>, dim3(grid), dim3(kBlock), 0, c->stream, frame->fv, first, step, n, (uint32_t)offset, \
                     (uint32_t)size, (const WlSlot*)wl->set.p, wl->capacity - 1, d_valid, d_call)
    if (size <= 16) FQG_WL(2);
    else if (size <= 32) FQG_WL(4);
    else FQG_WL(7);
#undef FQG_WL
  }
  WlCall* h_call = reinterpret_cast<WlCall*>(c->h_bcall);
  HIP_TRY(c, hipMemcpyAsync(h_call, d_call, sizeof(WlCall), hipMemcpyDeviceToHost, c->stream));
  if (valid) HIP_TRY(c, hipMemcpyAsync(valid, d_valid, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  out->n_valid = h_call->n_valid;
  out->n_short = h_call->n_short;
  return 0;
}

// ---- per-record filters ---------------------------------------------------------------------
int fqg_records_filter(fqg_ctx* c, const fqg_frame* frame, uint64_t first_record, uint64_t n_rec,
                       const fqg_filter_params* fp, fqg_filter_result* out) {
  if (!c || !frame || !fp || !out) return FQG_ERR_ARG;
  if (c->out_pending) {  // (a copy of the previous output is still on its way: this call writes the same buffers)
    const int rcw = fqg_barcodes_output_wait(c);
    if (rcw) return rcw;
  }
  memset(out, 0, sizeof(*out));
  c->bc_out_bytes[0] = c->bc_out_bytes[1] = c->bc_out_bytes[2] = 0;
  if (fp->mode != FQG_FILTER_N && fp->mode != FQG_FILTER_POLY_AT) return fail(c, FQG_ERR_ARG, "fqg_records_filter: unknown mode");
  if (first_record + n_rec > frame->fv.n_records) return fail(c, FQG_ERR_ARG, "fqg_records_filter: records beyond the frame");
  HIP_TRY(c, hipSetDevice(c->device));
  out->n_records = n_rec;
  if (!n_rec) return 0;
  BcParams F;
  memset(&F, 0, sizeof(F));
  F.f[1].fv = frame->fv;
  F.f[1].first = first_record;
  F.f[1].step = 1;
  F.f[1].present = 1;
  F.f[1].has_nul = F.has_nul = (frame->flags & kFlagNul) ? 1 : 0;  // lines are C strings then (bc_clip_nul)
  F.n_inputs = 1;
  F.emit[1] = 1;
  RfParams P;
  P.mode = fp->mode;
  P.max_n = fp->max_n_percent > 100 ? 100 : fp->max_n_percent;
  P.min_poly = fp->min_poly_at_len;
  P.min_len = (uint64_t)fp->min_len;
  int rc;
  const uint64_t nb = (n_rec + kScan64Span - 1) / kScan64Span;
  if ((rc = ensure(c, c->bc_status, n_rec))) return rc;
  if ((rc = ensure(c, c->bc_len[1], n_rec * 4))) return rc;
  if ((rc = ensure(c, c->bc_off[1], n_rec * 8))) return rc;
  if ((rc = ensure(c, c->bc_sum[1], nb * 8))) return rc;
  BcCall z;
  memset(&z, 0, sizeof(z));
  *c->h_bcall = z;
  HIP_TRY(c, hipMemcpyAsync(c->d_bcall, c->h_bcall, sizeof(BcCall), hipMemcpyHostToDevice, c->stream));
  // (the tiles are the EMIT kernel's alone now - the plan works record by record - and that kernel likes 24 KiB per
  // wavefront better than the 20 KiB it shared with a staging plan: tools/tiles_lds_sweep.sh)
  BcTile tc = bc_tile_for(F, 24576u);
  tc.plan_m = 1;
  const uint64_t n_tiles = (n_rec + tc.T - 1) / tc.T;
  if ((rc = ensure(c, c->bc_tile_big, n_tiles))) return rc;
  auto resident = [&](const void* kernel, unsigned lds) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kWave, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    return (unsigned)per_cu * (unsigned)c->cu_count;
  };
  unsigned long long* d_tot = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->d_bcall) + sizeof(BcCall));
  unsigned long long* h_tot = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(c->h_bcall) + sizeof(BcCall));
  {
    ProfScope ps(c, "k_rf_plan");
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_rec + kBlock - 1) / kBlock, (uint64_t)c->cu_count * 32));
    if (P.mode == FQG_FILTER_N) {
      const unsigned grid_n = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_rec + 31) / 32, (uint64_t)c->cu_count * 32));
      hipLaunchKernelGGL(k_rf_plan_n, dim3(grid_n), dim3(kBlock), 0, c->stream, F, P, n_rec, (uint8_t*)c->bc_status.p,
                         (uint32_t*)c->bc_len[1].p, c->d_bcall);
    } else {
      hipLaunchKernelGGL(k_rf_plan_records, dim3(grid), dim3(kBlock), 0, c->stream, F, P, n_rec, (uint8_t*)c->bc_status.p,
                         (uint32_t*)c->bc_len[1].p, c->d_bcall);
    }
  }
  {
    // (the plan kernels count what they drop and trim; the tile flags come behind the scan, whose offsets they use)
    ProfScope ps(c, "k_rf_scan");
    hipLaunchKernelGGL(k_scan64_a, dim3((unsigned)nb), dim3(kBlock), 0, c->stream, (const uint32_t*)c->bc_len[1].p, n_rec,
                       (unsigned long long*)c->bc_off[1].p, (unsigned long long*)c->bc_sum[1].p);
    hipLaunchKernelGGL(k_scan64_b, dim3(1), dim3(kBlock), 0, c->stream, (unsigned long long*)c->bc_sum[1].p, nb, d_tot + 1);
    hipLaunchKernelGGL(k_rf_tile_flags, dim3((unsigned)((n_tiles + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, F, tc, n_rec,
                       (const uint32_t*)c->bc_len[1].p, (const unsigned long long*)c->bc_off[1].p,
                       (const unsigned long long*)c->bc_sum[1].p, (uint8_t*)c->bc_tile_big.p, c->d_bcall);
  }
  HIP_TRY(c, hipMemcpyAsync(c->h_bcall, c->d_bcall, sizeof(BcCall) + 64, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  out->n_discarded = c->h_bcall->discarded;
  out->n_trimmed = c->h_bcall->short_warnings;
  out->n_kept = n_rec - out->n_discarded;
  out->out_bytes = h_tot[1];
  c->bc_out_bytes[1] = h_tot[1];
  if ((rc = ensure(c, c->bc_out[1], std::max<uint64_t>(h_tot[1], 16)))) return rc;
  if (h_tot[1]) {
    ProfScope ps(c, "k_rf_emit");
    const EmitOut eo{(const uint32_t*)c->bc_len[1].p, (const unsigned long long*)c->bc_off[1].p,
                     (const unsigned long long*)c->bc_sum[1].p, (uint8_t*)c->bc_out[1].p};
    const unsigned lds = tc.in_cap + tc.out_cap;
    const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, resident((const void*)k_rf_emit_tile, lds));
    hipLaunchKernelGGL(k_rf_emit_tile, dim3(grid), dim3(kWave), lds, c->stream, F, P, tc, n_rec,
                       (const uint8_t*)c->bc_status.p, (const uint8_t*)c->bc_tile_big.p, eo);
    if (c->h_bcall->big) {
      const unsigned grid_e = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n_rec + 3) / 4, (uint64_t)c->cu_count * 8));
      hipLaunchKernelGGL(k_rf_emit_direct, dim3(grid_e), dim3(kBlock), 0, c->stream, F, P, tc, n_rec,
                         (const uint8_t*)c->bc_status.p, (const uint8_t*)c->bc_tile_big.p, eo);
    }
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  return 0;
}

int fqg_barcodes_output(fqg_ctx* c, int which, void* host_dst, uint64_t nbytes) {
  if (!c || which < 0 || which > 2 || (!host_dst && nbytes)) return FQG_ERR_ARG;
  if (nbytes > c->bc_out_bytes[which]) return fail(c, FQG_ERR_ARG, "fqg_barcodes_output: more than was produced");
  if (!nbytes) return 0;
  HIP_TRY(c, hipMemcpyAsync(host_dst, c->bc_out[which].p, nbytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return 0;
}

int fqg_barcodes_output_wait(fqg_ctx* c) {
  if (!c) return FQG_ERR_ARG;
  if (!c->out_pending) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->out_stream));
  c->out_pending = false;
  return 0;
}

int fqg_barcodes_output_begin(fqg_ctx* c, int which, void* host_dst, uint64_t nbytes) {
  if (!c || which < 0 || which > 2 || (!host_dst && nbytes)) return FQG_ERR_ARG;
  if (nbytes > c->bc_out_bytes[which]) return fail(c, FQG_ERR_ARG, "fqg_barcodes_output_begin: more than was produced");
  if (!nbytes) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->out_stream) {
    HIP_TRY(c, hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->out_ready, hipEventDisableTiming));
  }
  HIP_TRY(c, hipEventRecord(c->out_ready, c->stream));  // (what produced the text has been launched on `stream`)
  HIP_TRY(c, hipStreamWaitEvent(c->out_stream, c->out_ready, 0));
  HIP_TRY(c, hipMemcpyAsync(host_dst, c->bc_out[which].p, nbytes, hipMemcpyDeviceToHost, c->out_stream));
  c->out_pending = true;
  return 0;
}

int fqg_records_filter_output(fqg_ctx* c, void* host_dst, uint64_t nbytes) { return fqg_barcodes_output(c, 1, host_dst, nbytes); }

int fqg_records_gather(fqg_ctx* c, const fqg_frame* frame, const uint64_t* records, uint64_t n, uint64_t* out_bytes) {
  if (!c || !frame || !out_bytes || (!records && n)) return FQG_ERR_ARG;
  if (c->out_pending) {  // (a copy of the previous output is still on its way: this call writes the same buffers)
    const int rcw = fqg_barcodes_output_wait(c);
    if (rcw) return rcw;
  }
  *out_bytes = 0;
  c->bc_out_bytes[0] = c->bc_out_bytes[1] = c->bc_out_bytes[2] = 0;
  if (!n) return 0;
  for (uint64_t k = 0; k < n; ++k)
    if (records[k] >= frame->fv.n_records) return fail(c, FQG_ERR_ARG, "fqg_records_gather: record outside the frame");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  const bool nul = (frame->flags & kFlagNul) != 0;
  const uint64_t nb = (n + kScan64Span - 1) / kScan64Span;
  if ((rc = ensure(c, c->bc_off[1], n * 8))) return rc;     // the list
  if ((rc = ensure(c, c->bc_len[1], n * 4))) return rc;
  if ((rc = ensure(c, c->bc_off[2], n * 8))) return rc;     // local offsets
  if ((rc = ensure(c, c->bc_sum[1], nb * 8 + 16))) return rc;
  unsigned long long* d_tot = (unsigned long long*)c->bc_sum[1].p + nb;
  HIP_TRY(c, hipMemcpyAsync(c->bc_off[1].p, records, n * 8, hipMemcpyHostToDevice, c->stream));
  {
    ProfScope ps(c, "k_gather_plan");
    if (nul)
      hipLaunchKernelGGL(k_gather_lens_nul, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, frame->fv,
                         (const unsigned long long*)c->bc_off[1].p, n, (uint32_t*)c->bc_len[1].p);
    else
      hipLaunchKernelGGL(k_gather_lens, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, frame->fv,
                         (const unsigned long long*)c->bc_off[1].p, n, (uint32_t*)c->bc_len[1].p);
    hipLaunchKernelGGL(k_scan64_a, dim3((unsigned)nb), dim3(kBlock), 0, c->stream, (const uint32_t*)c->bc_len[1].p, n,
                       (unsigned long long*)c->bc_off[2].p, (unsigned long long*)c->bc_sum[1].p);
    hipLaunchKernelGGL(k_scan64_b, dim3(1), dim3(kBlock), 0, c->stream, (unsigned long long*)c->bc_sum[1].p, nb, d_tot);
  }
  HIP_TRY(c, hipMemcpyAsync(&c->h_scalar[2], d_tot, 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const uint64_t total = c->h_scalar[2];
  if ((rc = ensure(c, c->bc_out[1], total + 64))) return rc;
  {
    ProfScope ps(c, "k_gather_copy");
    const unsigned grid = (unsigned)std::min<uint64_t>((n + 4 * kWave - 1) / (4 * kWave), (uint64_t)c->cu_count * 16);
    if (nul)  // what gzputs writes of a line is the C string: four pieces per record
      hipLaunchKernelGGL(k_gather_copy_nul, dim3((unsigned)std::min<uint64_t>((n + 3) / 4, (uint64_t)c->cu_count * 16)), dim3(kBlock), 0,
                         c->stream, frame->fv, (const unsigned long long*)c->bc_off[1].p, n,
                         (const unsigned long long*)c->bc_off[2].p, (const unsigned long long*)c->bc_sum[1].p,
                         (uint8_t*)c->bc_out[1].p);
    else
      hipLaunchKernelGGL(k_gather_copy, dim3(grid), dim3(kBlock), 0, c->stream, frame->fv,
                         (const unsigned long long*)c->bc_off[1].p, n, (const unsigned long long*)c->bc_off[2].p,
                         (const unsigned long long*)c->bc_sum[1].p, (const uint32_t*)c->bc_len[1].p, (uint8_t*)c->bc_out[1].p);
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  c->bc_out_bytes[1] = total;
  *out_bytes = total;
  return 0;
}
int fqg_records_gather_output(fqg_ctx* c, void* host_dst, uint64_t nbytes) { return fqg_barcodes_output(c, 1, host_dst, nbytes); }

uint64_t fqg_index_names_captured(const fqg_ctx* c) { return c ? c->last_names_captured : 0; }
const fqg_frame* fqg_index_frame(const fqg_index* ix, uint64_t k) {
  return (ix && k < ix->frames.size()) ? ix->frames[k] : nullptr;
}
uint64_t fqg_index_n_frames(const fqg_index* ix) { return ix ? ix->frames.size() : 0; }

// ---- measurement ----------------------------------------------------------------------------
int fqg_profile_enable(fqg_ctx* c, int on) {
  if (!c) return FQG_ERR_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_drain(c);
  c->profiling = on != 0;
  return 0;
}
int fqg_profile_reset(fqg_ctx* c) {
  if (!c) return FQG_ERR_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_drain(c);
  for (auto& s : c->slots) {
    s.launches = 0;
    s.ms = 0;
  }
  return 0;
}
int fqg_profile_read(fqg_ctx* c, fqg_kernel_time* out, size_t cap, size_t* n) {
  if (!c || !n) return FQG_ERR_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_drain(c);
  size_t k = 0;
  for (auto& s : c->slots) {
    if (k < cap && out) {
      memset(&out[k], 0, sizeof(out[k]));
      strncpy(out[k].name, s.name.c_str(), sizeof(out[k].name) - 1);
      out[k].launches = s.launches;
      out[k].total_ms = s.ms;
    }
    ++k;
  }
  *n = k;
  return 0;
}

// ---- synthetic data -------------------------------------------------------------------------
uint64_t fqg_synth_record_bytes(uint32_t read_len) { return (uint64_t)kSynthHdr + 2ull * (read_len + 1) + 2; }

int fqg_synth_fastq(fqg_ctx* c, void* device_out, uint64_t n_records, uint32_t read_len, uint64_t first_index,
                    uint64_t seed, int mate) {
  if (!c || !device_out || read_len == 0 || read_len >= FQG_MAX_READ_LENGTH - 2) return FQG_ERR_ARG;
  if (((uintptr_t)device_out & 15u) != 0) return fail(c, FQG_ERR_ARG, "output must be 16-byte aligned");
  if (first_index + n_records > 9999999999ull) return fail(c, FQG_ERR_ARG, "index does not fit 10 digits");
  if (!n_records) return 0;
  HIP_TRY(c, hipSetDevice(c->device));
  const uint64_t total = n_records * fqg_synth_record_bytes(read_len);
  const uint64_t threads = (total + 15) / 16;
  const uint64_t blocks = std::min<uint64_t>((threads + kBlock - 1) / kBlock, (uint64_t)c->cu_count * 32);
  ProfScope ps(c, "k_synth");
  hipLaunchKernelGGL(k_synth, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream, (uint8_t*)device_out, n_records,
                     read_len, first_index, seed, mate == 2 ? 2 : 1);
  HIP_TRY(c, hipGetLastError());
  return 0;
}

#include "fqg_umi_abi.inc"
#include "fqg_fp_abi.inc"
#include "fqg_bamtags_abi.inc"

}  // extern "C"
