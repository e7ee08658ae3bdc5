/* Test infrastructure (LD_PRELOAD for stress runs): every process's stderr goes to $VPF_STDERR_DIR/stderr_<pid>.txt (xdist does not pass its workers'
 * stderr on), and a SIGABRT / SIGSEGV / SIGBUS handler writes the NATIVE call stack there before the process dies — Python's faulthandler shows
 * which test line the main thread was in, not which library aborted, nor what the runtime printed before it did. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void on_fatal(int sig) {
  void* frames[64];
  const char msg[] = "\n[abort_backtrace] native stack of the signalled thread:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, backtrace(frames, 64), 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void install(void) {
  const char* dir = getenv("VPF_STDERR_DIR");
  if (dir) {
    char path[512];
    snprintf(path, sizeof(path), "%s/stderr_%d.txt", dir, (int)getpid());
    const int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) { dup2(fd, 2); close(fd); }
  }
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = on_fatal;
  sa.sa_flags = SA_NODEFER | SA_RESETHAND;
  sigaction(SIGABRT, &sa, 0);
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}
