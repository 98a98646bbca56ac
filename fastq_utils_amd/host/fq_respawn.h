// fq_respawn.h - starting a program over on input that must be cut at the reference's gzgets limits (fq_reframe.h).
//
// A plain file goes to the GPU as it is, in pieces read by many threads.  When the GPU reports a line beyond the
// limits (FQG_E_LINE_TOO_LONG: real files have none), the records BEFORE that line have been handled exactly as the
// reference handles them, and whatever the program has printed so far is a prefix of what the reference prints.  The
// program then runs itself again as a CHILD (never exec: this process holds the GPU) with
//
//     FQGPU_REFRAME=1         every input is cut while it is read (fq_input.h)
//     FQGPU_SKIP_OUT / _ERR   bytes of stdout / stderr that have been written already: the child drops as many
//
// and leaves with the child's status.  Output files are simply written again by the child.  The byte counts come from
// a pair of stdio streams installed over fd 1 and fd 2 at the start of main() (glibc lets stdout / stderr be
// assigned), so printf, fputs, fwrite and the message macros are all counted without touching a call site.
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern char** environ;

namespace fqhost {

struct CountedStream {
  int fd = -1;
  unsigned long long written = 0, skip = 0;  // bytes the program produced / bytes still to drop
};
inline CountedStream& counted(int which) {
  static CountedStream s[2];
  return s[which];
}
inline char**& saved_argv() {
  static char** v = nullptr;
  return v;
}
// set by the programs that reproduce the reference's cuts (install_counted_output): fastq_info.  The programs that COPY
// records to their outputs (fastq_filterpair, fastq_filter_n, fastq_trim_poly_at, fastq_pre_barcodes) do not - a cut image
// is not what they must write - and keep refusing lines beyond the limits.
inline bool& reframe_supported() {
  static bool v = false;
  return v;
}

// Where diagnostics go (FQGPU_TIMING, FQGPU_PGZIP_DEBUG): fd 2 itself, NOT the counted stream.  Their text carries
// times and differs from run to run, so it must neither be counted in the parent (the child's text of the same lines
// has another length: the skip would cut the child's real output in the wrong place) nor be dropped in the child.
inline FILE* diag() {
  static FILE* f = [] {
    FILE* g = fdopen(dup(2), "w");
    if (g) setvbuf(g, nullptr, _IONBF, 0);
    return g;
  }();
  return f ? f : stderr;
}

inline ssize_t counted_write(void* cookie, const char* buf, size_t n) {
  CountedStream* c = static_cast<CountedStream*>(cookie);
  c->written += n;
  size_t off = 0;
  if (c->skip) {
    off = (size_t)std::min<unsigned long long>(c->skip, n);
    c->skip -= off;
  }
  while (off < n) {
    const ssize_t w = write(c->fd, buf + off, n - off);
    if (w < 0) return 0;  // (stdio reports the error to the caller)
    off += (size_t)w;
  }
  return (ssize_t)n;
}

// first thing in main(): count what goes to stdout / stderr, and drop what a parent has written already
inline void install_counted_output(char** argv) {
  saved_argv() = argv;
  reframe_supported() = true;
  const char* names[2] = {"FQGPU_SKIP_OUT", "FQGPU_SKIP_ERR"};
  FILE** std_streams[2] = {&stdout, &stderr};
  for (int w = 0; w < 2; ++w) {
    CountedStream& c = counted(w);
    c.fd = w + 1;
    if (const char* e = getenv(names[w])) c.skip = strtoull(e, nullptr, 10);
    cookie_io_functions_t io;
    memset(&io, 0, sizeof(io));
    io.write = counted_write;
    FILE* f = fopencookie(&c, "w", io);
    if (!f) continue;
    // stderr unbuffered as it always is; stdout as stdio would have it
    if (w == 1) setvbuf(f, nullptr, _IONBF, 0);
    else setvbuf(f, nullptr, isatty(1) ? _IOLBF : _IOFBF, 1 << 16);
    *std_streams[w] = f;
  }
}

inline bool reframing() { return reframe_supported() && getenv("FQGPU_REFRAME") != nullptr; }

// the program again, on input cut at the gzgets limits; does not return
[[noreturn]] inline void respawn_reframed() {
  fflush(stdout);
  fflush(stderr);
  char** argv = saved_argv();
  if (reframing() || !argv) {  // (cannot be: a re-framed image has no line beyond the limits)
    const char msg[] = "\nERROR: internal: a line beyond the gzgets limits in re-framed input\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    _exit(2);
  }
  std::vector<std::string> keep;
  for (char** e = environ; *e; ++e)
    if (strncmp(*e, "FQGPU_SKIP_", 11) != 0 && strncmp(*e, "FQGPU_REFRAME=", 14) != 0 && strncmp(*e, "FQGPU_DEVICES=", 14) != 0)
      keep.emplace_back(*e);
  keep.emplace_back("FQGPU_REFRAME=1");  // (and one device: the cut is a serial walk over the file)
  keep.emplace_back("FQGPU_SKIP_OUT=" + std::to_string(counted(0).written));
  keep.emplace_back("FQGPU_SKIP_ERR=" + std::to_string(counted(1).written));
  std::vector<char*> envp;
  for (auto& s : keep) envp.push_back(const_cast<char*>(s.c_str()));
  envp.push_back(nullptr);
  pid_t pid = 0;
  if (posix_spawn(&pid, "/proc/self/exe", nullptr, nullptr, argv, envp.data()) != 0) {
    const char msg[] = "\nERROR: unable to start the program again for input beyond the line limits\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    _exit(2);
  }
  int status = 0;
  while (waitpid(pid, &status, 0) < 0) {
  }
  _exit(WIFEXITED(status) ? WEXITSTATUS(status) : 2);
}

}  // namespace fqhost
