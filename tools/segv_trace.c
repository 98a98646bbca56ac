/* LD_PRELOAD helper of tools/exit_stress.py: a process that dies of SIGSEGV / SIGBUS / SIGABRT while it LEAVES (exit()'s
 * hooks, static destructors, the HIP runtime's teardown) says where - a backtrace on stderr - and ends with the
 * signal's conventional status.   gcc -O1 -g -shared -fPIC -o segv_trace.so segv_trace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_signal(int sig, siginfo_t* si, void* ctx) {
  (void)ctx;
  char line[160];
  int n = snprintf(line, sizeof(line), "\n[segv_trace] signal %d, fault address %p, pid %d\n", sig, si ? si->si_addr : 0, (int)getpid());
  if (n > 0) (void)!write(2, line, (size_t)n);
  void* frames[64];
  const int depth = backtrace(frames, 64);
  backtrace_symbols_fd(frames, depth, 2);
  _exit(128 + sig);
}

/* SIGUSR1 (tools/repro_campaign_case.py sends it to every thread of a run that does not end): where this thread is */
static void on_ask(int sig) {
  (void)sig;
  char line[96];
  int n = snprintf(line, sizeof(line), "\n[segv_trace] thread %d of pid %d is here:\n", (int)gettid(), (int)getpid());
  if (n > 0) (void)!write(2, line, (size_t)n);
  void* frames[64];
  const int depth = backtrace(frames, 64);
  backtrace_symbols_fd(frames, depth, 2);
}

__attribute__((constructor)) static void install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = on_signal;
  sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
  sigaction(SIGABRT, &sa, 0);
  struct sigaction ask;
  memset(&ask, 0, sizeof(ask));
  ask.sa_handler = on_ask;
  ask.sa_flags = SA_RESTART;
  sigaction(SIGUSR1, &ask, 0);
}
