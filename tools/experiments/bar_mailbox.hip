// tools/experiments/bar_mailbox.hip -- can the host write a request straight into DEVICE memory (large BAR), and what does a
// round trip through a resident wavefront cost that way against a request word in mapped host memory? (the small-batch server's
// floor: DESIGN 4.6)   hipcc --offload-arch=gfx950 -O2 -o /tmp/bar_mailbox tools/experiments/bar_mailbox.hip && /tmp/bar_mailbox
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <setjmp.h>
#include <signal.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                                   \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

// one wavefront: waits for *req to change, sums `words` dwords of the request's payload, answers into host memory
__global__ void echo_kernel(volatile uint32_t *req, const uint32_t *payload, uint32_t words, volatile uint32_t *done, volatile uint32_t *sum_out,
                            uint32_t rounds) {
    uint32_t last = 0;
    const unsigned long long t_begin = wall_clock64(); /* 100 MHz; the kernel gives up after 5 s whatever happens */
    for (uint32_t r = 0; r < rounds; r++) {
        uint32_t seq;
        do {
            seq = __hip_atomic_load((uint32_t *)req, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            seq = __builtin_amdgcn_readfirstlane(seq);
            if (wall_clock64() - t_begin > 500000000ull) return;
        } while (seq == last);
        last = seq;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        uint32_t s = 0;
        for (uint32_t i = threadIdx.x; i < words; i += 64) s += payload[i]; /* (ordinary loads: what the scan's body uses) */
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
        if (threadIdx.x == 0) *sum_out = s;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (threadIdx.x == 0) __hip_atomic_store((uint32_t *)done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static sigjmp_buf g_jmp;
static void on_fault(int) { siglongjmp(g_jmp, 1); }
static int try_host_access(const char *what, uint32_t *p) {
    struct sigaction sa, o1, o2;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_fault;
    sigaction(SIGSEGV, &sa, &o1);
    sigaction(SIGBUS, &sa, &o2);
    int ok = 0;
    if (sigsetjmp(g_jmp, 1) == 0) {
        volatile uint32_t *v = p;
        v[0] = 0x1234;
        v[1] = v[0] + 1;
        ok = v[1] == 0x1235 ? 1 : -1;
    }
    sigaction(SIGSEGV, &o1, nullptr);
    sigaction(SIGBUS, &o2, nullptr);
    printf("host store/load through a %s pointer: %s\n", what, ok == 1 ? "works" : ok == 0 ? "FAULT" : "wrong value");
    return ok == 1;
}

static int pingpong(const char *what, uint32_t *req, uint32_t *d_req, uint32_t *payload, uint32_t *d_payload, uint32_t bytes, int rounds) {
    uint32_t *done, *sum;
    CK(hipHostMalloc((void **)&done, 4096, hipHostMallocMapped));
    sum = done + 64;
    done[0] = 0;
    uint32_t *d_done, *d_sum;
    CK(hipHostGetDevicePointer((void **)&d_done, done, 0));
    d_sum = d_done + 64;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(64), 0, st, (volatile uint32_t *)d_req, (const uint32_t *)d_payload, bytes / 4, (volatile uint32_t *)d_done,
                       (volatile uint32_t *)d_sum, (uint32_t)rounds);
    CK(hipGetLastError());
    uint32_t buf[512];
    for (int i = 0; i < 512; i++) buf[i] = i;
    double best = 1e9, total = 0;
    int bad = 0;
    for (int r = 1; r <= rounds; r++) {
        for (uint32_t i = 0; i < bytes / 4; i++) buf[i] = i * 2654435761u + (uint32_t)r * 40503u; /* every word new every round: a stale line shows */
        auto t0 = std::chrono::steady_clock::now();
        memcpy(payload, buf, bytes);
        _mm_sfence(); /* (device memory through the BAR is write-combining: without the fences the stores sit in the CPU's buffers) */
        __atomic_store_n(req, (uint32_t)r, __ATOMIC_RELEASE);
        _mm_sfence();
        while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != (uint32_t)r) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
                printf("%s: no answer to request %d within 2 s\n", what, r);
                (void)hipStreamSynchronize(st); /* (the kernel gives up by itself) */
                return 0;
            }
        }
        auto t1 = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        uint32_t want = 0;
        for (uint32_t i = 0; i < bytes / 4; i++) want += buf[i];
        if (*sum != want) bad++;
        if (r > 20) total += us, best = us < best ? us : best;
    }
    CK(hipStreamSynchronize(st));
    printf("%-44s %4u-byte request: %.2f us per round trip (best %.2f), %d wrong sums of %d\n", what, bytes, total / (rounds - 20), best, bad, rounds);
    CK(hipStreamDestroy(st));
    CK(hipHostFree(done));
    return 0;
}

int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    printf("%s: isLargeBar %d\n", pr.name, pr.isLargeBar);
    uint32_t *d_plain = nullptr, *d_fine = nullptr, *h_map = nullptr, *d_map = nullptr;
    CK(hipMalloc((void **)&d_plain, 1 << 16));
    hipError_t e = hipExtMallocWithFlags((void **)&d_fine, 1 << 16, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    CK(hipHostMalloc((void **)&h_map, 1 << 16, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&d_map, h_map, 0));
    CK(hipMemset(d_plain, 0, 1 << 16));
    if (e == hipSuccess) CK(hipMemset(d_fine, 0, 1 << 16));
    memset(h_map, 0, 1 << 16);
    CK(hipDeviceSynchronize());
    const int plain_ok = try_host_access("hipMalloc", d_plain);
    const int fine_ok = e == hipSuccess ? try_host_access("fine-grained device", d_fine) : 0;
    for (uint32_t bytes : {64u, 1472u}) {
        // the request word at [0], the payload behind it (another cache line)
        if (pingpong("request + payload in mapped HOST memory", h_map, d_map, h_map + 64, d_map + 64, bytes, 2000)) return 1;
        if (fine_ok) {
            CK(hipMemset(d_fine, 0, 1 << 16));
            CK(hipDeviceSynchronize());
            if (pingpong("request + payload in fine-grained DEVICE memory", d_fine, d_fine, d_fine + 64, d_fine + 64, bytes, 2000)) return 1;
        }
        if (plain_ok) {
            CK(hipMemset(d_plain, 0, 1 << 16));
            CK(hipDeviceSynchronize());
            if (pingpong("request + payload in hipMalloc memory", d_plain, d_plain, d_plain + 64, d_plain + 64, bytes, 2000)) return 1;
        }
    }
    return 0;
}
