/* native_bt.c -- test infrastructure: print the NATIVE stack and the name of the thread that raised a fatal signal.
 *
 * Python's faulthandler shows Python frames only; round 2's one unexplained `Fatal Python error: Aborted`
 * (gpurun_out/r2_t13.log) came from a thread that has no Python state at all (no "Current thread" in its dump), i.e. from a
 * runtime / library worker thread whose stack was never seen.  tests/conftest.py installs this handler for SIGABRT, SIGSEGV
 * and SIGBUS in front of faulthandler's: if anything in a GPU test run ever aborts again, the log names the thread
 * (/proc/self/task/<tid>/comm) and the frames that called abort().  Chains to the handler that was installed before it. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static struct sigaction g_prev[3];
static const int g_sigs[3] = {SIGABRT, SIGSEGV, SIGBUS};

static void put(const char *s) { (void)!write(2, s, strlen(s)); }
static void put_int(long v) {
  char buf[24];
  int n = 0;
  if (v == 0) buf[n++] = '0';
  while (v > 0 && n < 23) {
    buf[n++] = (char)('0' + v % 10);
    v /= 10;
  }
  while (n > 0) (void)!write(2, &buf[--n], 1);
}

static void on_fatal(int sig, siginfo_t *si, void *ctx) {
  static volatile sig_atomic_t busy = 0;
  if (!busy) {
    busy = 1;
    const long tid = syscall(SYS_gettid);
    put("\n=== native_bt: signal ");
    put_int(sig);
    put(" raised on thread ");
    put_int(tid);
    char path[64] = "/proc/self/task/", comm[64];
    /* append tid */
    {
      char num[24];
      int n = 0;
      long v = tid;
      while (v > 0 && n < 23) {
        num[n++] = (char)('0' + v % 10);
        v /= 10;
      }
      size_t len = strlen(path);
      while (n > 0) path[len++] = num[--n];
      path[len] = 0;
      strcat(path, "/comm");
    }
    int fd = open(path, O_RDONLY);
    if (fd >= 0) {
      ssize_t k = read(fd, comm, sizeof(comm) - 1);
      close(fd);
      if (k > 0) {
        comm[k] = 0;
        put(" (");
        if (comm[k - 1] == '\n') comm[k - 1] = 0;
        put(comm);
        put(")");
      }
    }
    put("; native frames:\n");
    void *frames[96];
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    put("=== end native_bt ===\n");
    busy = 0;
  }
  for (int i = 0; i < 3; i++)
    if (g_sigs[i] == sig) {
      struct sigaction *p = &g_prev[i];
      if ((p->sa_flags & SA_SIGINFO) && p->sa_sigaction) {
        p->sa_sigaction(sig, si, ctx);
        return;
      }
      if (!(p->sa_flags & SA_SIGINFO) && p->sa_handler != SIG_DFL && p->sa_handler != SIG_IGN) {
        p->sa_handler(sig);
        return;
      }
    }
  signal(sig, SIG_DFL);
  raise(sig);
}

int native_bt_install(void) {
  void *warm[4];
  (void)backtrace(warm, 4); /* loads libgcc now: not from inside a signal handler */
  for (int i = 0; i < 3; i++) {
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_fatal;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    if (sigaction(g_sigs[i], &sa, &g_prev[i]) != 0) return -1;
  }
  return 0;
}
