// How much host CPU does each way of waiting for a busy stream cost on this runtime?
// Build: hipcc --offload-arch=gfx950 -O2 -o waitprobe waitprobe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <vector>
static double thread_cpu() { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double wall() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void spin(unsigned long long cycles, unsigned *out)
{
	unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < cycles) { }
	if (out) *out = 1;
}
int main(int argc, char **argv)
{
	int mode_flags = argc > 1 ? atoi(argv[1]) : 0;
	if (mode_flags == 1) hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
	if (mode_flags == 2) hipSetDeviceFlags(hipDeviceScheduleYield);
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	unsigned *d; hipMalloc(&d, 4096);
	unsigned *pinned; hipHostMalloc(&pinned, 4096);
	unsigned pageable[16];
	hipEvent_t evb, evn; hipEventCreateWithFlags(&evb, hipEventBlockingSync | hipEventDisableTiming); hipEventCreateWithFlags(&evn, hipEventDisableTiming);
	const unsigned long long cyc = 30000000ull; // 100 MHz wall clock -> 300 ms
	spin<<<1, 64, 0, s>>>(1000, d); hipStreamSynchronize(s);
	const char *names[] = {"hipStreamSynchronize", "blocking event sync", "plain event sync", "memcpyAsync D2H pageable (+sync)", "memcpyAsync D2H pinned + blocking event", "hipStreamQuery poll + 200us nanosleep", "blocking event sync, then pageable D2H"};
	for (int m = 0; m < 7; m++) {
		double c0 = thread_cpu(), w0 = wall();
		spin<<<1, 64, 0, s>>>(cyc, d);
		switch (m) {
		case 0: hipStreamSynchronize(s); break;
		case 1: hipEventRecord(evb, s); hipEventSynchronize(evb); break;
		case 2: hipEventRecord(evn, s); hipEventSynchronize(evn); break;
		case 3: hipMemcpyAsync(pageable, d, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); break;
		case 4: hipMemcpyAsync(pinned, d, 4, hipMemcpyDeviceToHost, s); hipEventRecord(evb, s); hipEventSynchronize(evb); break;
		case 5: { timespec ts{0, 200000}; while (hipStreamQuery(s) == hipErrorNotReady) nanosleep(&ts, nullptr); } break;
		case 6: hipEventRecord(evb, s); hipEventSynchronize(evb); hipMemcpyAsync(pageable, d, 4, hipMemcpyDeviceToHost, s); hipEventRecord(evb, s); hipEventSynchronize(evb); break;
		}
		printf("flags=%d %-44s wall %.1f ms  thread cpu %.1f ms\n", mode_flags, names[m], 1e3 * (wall() - w0), 1e3 * (thread_cpu() - c0));
	}
	return 0;
}
