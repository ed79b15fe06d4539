// tools/experiments/vmm_semantics.hip -- what hipMemcpy* / hipMemset* do on ranges mapped with the virtual-memory API (hipMemCreate +
// hipMemMap), against the same calls on hipMalloc memory. Built and run by hand: hipcc --offload-arch=gfx950 -O2 vmm_semantics.hip -o vmm && ./vmm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void slow_store(unsigned *p, unsigned v, int spin) {
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) { }
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) p[i] = v;
}
__global__ void sum_k(const unsigned *p, unsigned long long *out, int n) {
    unsigned long long s = 0;
    for (int i = 0; i < n; i++) s += p[i];
    *out = s;
}

static int vmm_alloc(void **out, size_t bytes) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    size_t sz = (bytes + gran - 1) / gran * gran;
    void *va; CK(hipMemAddressReserve(&va, sz, gran, nullptr, 0));
    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, sz, &prop, 0));
    CK(hipMemMap(va, sz, 0, h, 0));
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, sz, &acc, 1));
    printf("granularity %zu\n", gran);
    *out = va; return 0;
}

static int run(const char *name, unsigned *d) {
    hipStream_t st; CK(hipStreamCreate(&st));  // blocking stream
    unsigned long long *d_sum; CK(hipMalloc(&d_sum, 8));
    std::vector<unsigned> h(1024);
    // 1: kernel on a blocking stream, then hipMemcpy D2H (null stream): must see the kernel's stores
    int stale = 0;
    for (int rep = 0; rep < 20; rep++) {
        CK(hipMemset(d, 0, 4096)); CK(hipDeviceSynchronize());
        slow_store<<<1, 64, 0, st>>>(d, 0xabcd0000u + rep, 200000);  // ~2 ms
        CK(hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost));
        if (h[5] != 0xabcd0000u + rep) stale++;
        CK(hipDeviceSynchronize());
    }
    printf("%s: hipMemcpy D2H after a kernel on a blocking stream: stale in %d of 20\n", name, stale);
    // 2: hipMemcpyAsync H2D from a pageable buffer that is overwritten right after the call returns
    int clobbered = 0;
    for (int rep = 0; rep < 20; rep++) {
        unsigned stack_buf[1024];
        for (int i = 0; i < 1024; i++) stack_buf[i] = 7;
        CK(hipMemcpyAsync(d, stack_buf, 4096, hipMemcpyHostToDevice, st));
        for (int i = 0; i < 1024; i++) stack_buf[i] = 9;  // legal for pageable memory under the CUDA/HIP staging contract
        sum_k<<<1, 1, 0, st>>>(d, d_sum, 1024);
        unsigned long long s = 0;
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
        if (s != 7 * 1024) clobbered++;
    }
    printf("%s: hipMemcpyAsync H2D from pageable memory reused after the call: wrong in %d of 20\n", name, clobbered);
    // 3: hipMemsetAsync then a kernel on the same stream
    int unordered = 0;
    for (int rep = 0; rep < 20; rep++) {
        slow_store<<<1, 64, 0, st>>>(d, 5, 100);
        CK(hipMemsetAsync(d, 0, 4096, st));
        sum_k<<<1, 1, 0, st>>>(d, d_sum, 1024);
        unsigned long long s = 1;
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
        if (s != 0) unordered++;
    }
    printf("%s: hipMemsetAsync between two kernels of one stream: out of order in %d of 20\n", name, unordered);
    // 4: synchronous hipMemcpy H2D then a kernel
    int h2d = 0;
    for (int rep = 0; rep < 20; rep++) {
        for (int i = 0; i < 1024; i++) h[i] = rep + 1;
        CK(hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice));
        sum_k<<<1, 1, 0, st>>>(d, d_sum, 1024);
        unsigned long long s = 0;
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
        if (s != (unsigned long long)(rep + 1) * 1024) h2d++;
    }
    printf("%s: hipMemcpy H2D then a kernel: wrong in %d of 20\n", name, h2d);
    hipPointerAttribute_t at{};
    hipError_t e = hipPointerGetAttributes(&at, d);
    printf("%s: hipPointerGetAttributes -> %s, type %d\n", name, hipGetErrorString(e), (int)at.type);
    return 0;
}

__global__ void fill_k(unsigned *p, unsigned v, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = v; }

// 5: free a mapping (unmap, release, free the address range) and map NEW physical memory -- the runtime hands the same address
// range out again --: do kernels see the new memory?
static int remap_test(bool free_va) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned long long *d_sum; CK(hipMalloc(&d_sum, 8));
    int same_va = 0, wrong = 0; void *prev = nullptr;
    for (int rep = 0; rep < 200; rep++) {
        size_t sz = gran * (1 + rep % 3);
        void *va; CK(hipMemAddressReserve(&va, sz + 2 * gran, gran, nullptr, 0));
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, sz, &prop, 0));
        void *m = (char *)va + gran;
        CK(hipMemMap(m, sz, 0, h, 0)); CK(hipMemSetAccess(m, sz, &acc, 1));
        if (va == prev) same_va++;
        prev = va;
        fill_k<<<1, 256>>>((unsigned *)m, 100 + rep, 1024);
        sum_k<<<1, 1>>>((const unsigned *)m, d_sum, 1024);
        unsigned long long s = 0; CK(hipMemcpy(&s, d_sum, 8, hipMemcpyDeviceToHost));
        if (s != (unsigned long long)(100 + rep) * 1024) wrong++;
        // a second allocation that stays: the victim of a stale translation, if there is one
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(m, sz)); CK(hipMemRelease(h));
        if (free_va) CK(hipMemAddressFree(va, sz + 2 * gran));
    }
    printf("remap (%s the address range): same address again %d of 200, kernel saw wrong data %d of 200\n", free_va ? "freeing" : "keeping", same_va, wrong);
    return 0;
}

int main() {
    unsigned *a, *b;
    CK(hipMalloc(&a, 1 << 20));
    void *v; if (vmm_alloc(&v, 1 << 20)) return 1;
    b = (unsigned *)v;
    if (run("hipMalloc", a)) return 1;
    if (run("vmm     ", b)) return 1;
    if (remap_test(true)) return 1;
    if (remap_test(false)) return 1;
    return 0;
}
