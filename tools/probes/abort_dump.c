/* Developer probe: LD_PRELOAD this into a pytest run whose process dies of abort() -- e.g. the ROCm runtime's handler
 * of a GPU memory fault, which prints the faulting address to stderr and aborts.  Under pytest's fd capture that text
 * lands in an unlinked temp file and is lost with the process; this abort() copies the tail of whatever fd 2 points at
 * to $ABORT_DUMP first.   gcc -shared -fPIC -O2 abort_dump.c -o abort_dump.so */
#define _GNU_SOURCE
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

void abort(void) {
	fflush(stderr);
	const char *path = getenv("ABORT_DUMP");
	int in = open("/proc/self/fd/2", O_RDONLY);
	int out = path ? open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644) : -1;
	if (in >= 0 && out >= 0) {
		off_t sz = lseek(in, 0, SEEK_END);
		off_t st = sz > 16384 ? sz - 16384 : 0;
		lseek(in, st, SEEK_SET);
		static char buf[16384];
		ssize_t n = read(in, buf, sizeof buf);
		if (n > 0) (void)!write(out, buf, (size_t)n);
		close(out);
	}
	signal(SIGABRT, SIG_DFL);
	raise(SIGABRT);
	_exit(134);
}
