// tools/ubench/lds_occupancy.hip — how many 64-thread workgroups with N KB of LDS does a CU of this chip keep resident?
// (the question behind k_hamt_lv_parse_actor's 16 KB per wavefront: 64 KB or 160 KB of LDS per CU for ordinary launches?)
// Every workgroup spins until `resident` stops growing, so the peak of the counter IS the number resident at once.
#include <hip/hip_runtime.h>

#include <cstdio>

template <int KB>
__global__ __launch_bounds__(64) void k_hold(unsigned* counter, unsigned* peak, unsigned long long cycles) {
    __shared__ unsigned char lds[KB * 1024];
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned now = atomicAdd(counter, 1u) + 1u;
        atomicMax(peak, now);
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < cycles) {
        }
        atomicSub(counter, 1u);
    }
    __syncthreads();
    if (lds[(threadIdx.x * 7) % (KB * 1024)] == 255 && counter == nullptr) printf("x");
}

template <int KB>
void run(unsigned* d, int cus) {
    hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k_hold<KB>, dim3(cus * 40), dim3(64), 0, 0, d, d + 1, 2000000ull);  // 20 ms of 100 MHz ticks … each holds ~20 ms? no: wall_clock64 = 100 MHz → 2e6 ticks = 20 ms
    hipDeviceSynchronize();
    unsigned h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_hold<KB>, 64, 0);
    std::printf("LDS %3d KB per 64-thread workgroup: peak resident %5u = %.2f per CU   (runtime's occupancy answer: %d per CU)\n", KB, h[1],
                double(h[1]) / cus, occ);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    std::printf("%s: %d CUs, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlockOptin %zu\n", p.gcnArchName,
                p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlockOptin);
    unsigned* d;
    hipMalloc(&d, 8);
    run<4>(d, p.multiProcessorCount);
    run<8>(d, p.multiProcessorCount);
    run<16>(d, p.multiProcessorCount);
    run<17>(d, p.multiProcessorCount);
    run<20>(d, p.multiProcessorCount);
    run<32>(d, p.multiProcessorCount);
    run<40>(d, p.multiProcessorCount);
    run<64>(d, p.multiProcessorCount);
    return 0;
}
