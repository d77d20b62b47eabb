// clock_probe.hip — what the shader clock is WHILE another stream keeps the GPU busy (the same
// compositing kernel took 0.2355 ms in the forward-only loop and 0.1996 ms inside fwd+bwd steps).
// One wave spins for a few microseconds and reports (core-clock cycles) / (constant-rate wall-clock ticks):
// clock64() counts shader cycles, wall_clock64() a constant-rate counter (hipDeviceAttributeWallClockRate kHz).
// Built as a tiny shared library driven through ctypes (tools/clock_experiment.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_clock_probe(unsigned long long *out, int slot, int spins) {
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long w1 = w0, c1 = c0;
    for (int i = 0; i < spins; ++i) {
        __builtin_amdgcn_s_sleep(8);
        w1 = wall_clock64(); c1 = clock64();
    }
    if (threadIdx.x == 0) { out[2 * slot] = c1 - c0; out[2 * slot + 1] = w1 - w0; }
}

extern "C" int clock_probe_launch(void *out, int slot, int spins, void *stream) {
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long *)out, slot, spins);
    return (int)hipGetLastError();
}
extern "C" int clock_probe_wall_khz(void) {
    int khz = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return 0;
    return khz;
}
