// poor man's perf: SIGPROF samples of RIP, dumped as a histogram of addresses
#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>
#include <cstdio>
#include <cstdlib>
#include <map>
static unsigned long long g_samples[1 << 20];
static unsigned g_n;
static void on_prof(int, siginfo_t *, void *uc)
{
	if (g_n < (1 << 20))
		g_samples[g_n++] = ((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP];
}
void sampler_start()
{
	struct sigaction sa = {};
	sa.sa_sigaction = on_prof;
	sa.sa_flags = SA_SIGINFO | SA_RESTART;
	sigaction(SIGPROF, &sa, nullptr);
	itimerval it = {{0, 200}, {0, 200}};
	setitimer(ITIMER_PROF, &it, nullptr);
}
void sampler_dump(const char *path)
{
	itimerval it = {{0, 0}, {0, 0}};
	setitimer(ITIMER_PROF, &it, nullptr);
	std::map<unsigned long long, unsigned> h;
	for (unsigned i = 0; i < g_n; i++)
		h[g_samples[i]]++;
	FILE *f = fopen(path, "w");
	for (auto &kv : h)
		fprintf(f, "%llx %u\n", kv.first, kv.second);
	fclose(f);
	fprintf(stderr, "%u samples\n", g_n);
}
