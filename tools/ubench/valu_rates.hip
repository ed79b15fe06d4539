// valu_rates.hip -- issue-rate microbenchmarks for the instruction mix of the literal-scan
// filter kernel on gfx950: cycles per wave64 instruction for the integer / SDWA / VOP3 forms it
// uses, and LDS random-access cost for ds_read_b32 / b64 / u8 (aligned and misaligned).
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 8 independent chains x 8 repeats = 64 instructions per loop trip
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define BODY(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)

#define KERNEL(NAME, ASMSTR)                                                                   \
    __global__ __launch_bounds__(1024) void k_##NAME(uint32_t *out, int trips, uint32_t seed, unsigned long long *cyc) { \
        uint32_t r0 = threadIdx.x * 2654435761u + seed, r1 = r0 ^ 0x1234567, r2 = r0 + 77, r3 = r0 * 3, r4 = r0 ^ 0xabcdef, \
                 r5 = r0 + 12345, r6 = r0 * 7, r7 = r0 ^ 0x55aa55;                              \
        uint32_t c = seed | 1, d = seed * 3 + 5;                                                \
        unsigned long long t0 = __builtin_readcyclecounter();                                 \
        for (int i = 0; i < trips; i++) {                                                       \
            asm volatile(ASMSTR : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(c), "v"(d) : "vcc"); \
        }                                                                                       \
        unsigned long long t1 = __builtin_readcyclecounter();                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;   \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                              \
    }

// each S(i) emits one instruction on chain i (operands %0..%7 chains, %8 = c, %9 = d)
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define I_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_SHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define I_SHRV(i) "v_lshrrev_b32 %" #i ", %8, %" #i "\n"
#define I_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define I_MULHI24(i) "v_mul_hi_u32_u24 %" #i ", %" #i ", %8\n"
#define I_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define I_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 5\n"
#define I_ALIGNBITV(i) "v_alignbit_b32 %" #i ", %" #i ", %8, %9\n"
#define I_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n"
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define I_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 7\n"
#define I_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define I_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define I_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define I_ADDSDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_ADDSDWA2(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_1\n"
#define I_SHRSDWA(i) "v_lshrrev_b32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define I_ANDSDWA(i) "v_and_b32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
#define I_MUL24SDWA(i) "v_mul_u32_u24_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
#define I_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_CMP(i) "v_cmp_lt_u32 vcc, %" #i ", %8\n"
#define I_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n"
#define I_PKMUL(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define I_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define I_PKSHR(i) "v_pk_lshrrev_b16 %" #i ", %8, %" #i "\n"
#define I_DOT2(i) "v_dot2_u32_u16 %" #i ", %" #i ", %8, %9\n"
#define I_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define I_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
#define I_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define I_FFBL(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define I_BPERM(i) "ds_bpermute_b32 %" #i ", %8, %" #i "\ns_waitcnt lgkmcnt(4)\n"

KERNEL(and, BODY(I_AND))
KERNEL(xor_, BODY(I_XOR))
KERNEL(add, BODY(I_ADD))
KERNEL(shr_imm, BODY(I_SHR))
KERNEL(shr_v, BODY(I_SHRV))
KERNEL(mul_u24, BODY(I_MUL24))
KERNEL(mul_hi_u24, BODY(I_MULHI24))
KERNEL(mad_u24, BODY(I_MAD24))
KERNEL(mul_lo_u32, BODY(I_MULLO))
KERNEL(alignbit_imm, BODY(I_ALIGNBIT))
KERNEL(alignbit_v, BODY(I_ALIGNBITV))
KERNEL(alignbyte, BODY(I_ALIGNBYTE))
KERNEL(perm, BODY(I_PERM))
KERNEL(bfe, BODY(I_BFE))
KERNEL(lshl_or, BODY(I_LSHLOR))
KERNEL(and_or, BODY(I_ANDOR))
KERNEL(or3, BODY(I_OR3))
KERNEL(add3, BODY(I_ADD3))
KERNEL(xad, BODY(I_XAD))
KERNEL(lshl_add, BODY(I_LSHLADD))
KERNEL(add_sdwa_b, BODY(I_ADDSDWA))
KERNEL(add_sdwa_bb, BODY(I_ADDSDWA2))
KERNEL(shr_sdwa, BODY(I_SHRSDWA))
KERNEL(and_sdwa, BODY(I_ANDSDWA))
KERNEL(mul24_sdwa, BODY(I_MUL24SDWA))
KERNEL(cndmask, BODY(I_CNDMASK))
KERNEL(cmp, BODY(I_CMP))
KERNEL(addc, BODY(I_ADDC))
KERNEL(pk_mul_lo_u16, BODY(I_PKMUL))
KERNEL(pk_add_u16, BODY(I_PKADD))
KERNEL(pk_lshrrev_b16, BODY(I_PKSHR))
KERNEL(dot2_u32_u16, BODY(I_DOT2))
KERNEL(dot4_u32_u8, BODY(I_DOT4))
KERNEL(fma_f32, BODY(I_FMA))
KERNEL(mov, BODY(I_MOV))
KERNEL(bfi, BODY(I_BFI))
KERNEL(mbcnt, BODY(I_MBCNT))
KERNEL(ffbl, BODY(I_FFBL))
KERNEL(bpermute, BODY(I_BPERM))

typedef void (*kfn)(uint32_t *, int, uint32_t, unsigned long long *);
struct Ent { const char *name; kfn f; };

// ---- LDS random access: MODE 0 b32 aligned, 1 b64 aligned, 2 u8, 3 b32 at odd byte offsets, 4 b64 at 4-byte
// (not 8-byte) aligned offsets, 5 b32 conflict-free (lane-own bank), 6 u16
template <int MODE>
__global__ __launch_bounds__(1024) void k_lds(uint32_t *out, int trips, uint32_t seed, unsigned long long *cyc) {
    extern __shared__ uint32_t lds[];
    const uint32_t nwords = 32768;
    for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a[8];
    uint32_t h = (threadIdx.x + blockIdx.x * 1024) * 2654435761u + seed;
    for (int j = 0; j < 8; j++) {
        h = h * 1664525u + 1013904223u;
        uint32_t byte = (h >> 8) % (nwords * 4 - 16);
        if (MODE == 0) byte &= ~3u;
        if (MODE == 1) byte &= ~7u;
        if (MODE == 3) byte |= 1u;
        if (MODE == 4) byte = (byte & ~7u) | 4u;
        if (MODE == 5) byte = ((byte >> 7) << 7) | ((threadIdx.x & 31) << 2);
        if (MODE == 6) byte &= ~1u;
        a[j] = byte;
    }
    uint32_t acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < trips; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            typedef __attribute__((address_space(3))) uint8_t *l8;
            l8 p = (l8)(uintptr_t)a[j];
            uint32_t v;
            if (MODE == 0 || MODE == 3 || MODE == 5) asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a[j]));
            else if (MODE == 1 || MODE == 4) { uint64_t w; asm volatile("ds_read_b64 %0, %1" : "=v"(w) : "v"(a[j])); v = (uint32_t)w ^ (uint32_t)(w >> 32); }
            else if (MODE == 2) asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(a[j]));
            else asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(a[j]));
            (void)p;
            asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
            acc ^= v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// correctness of misaligned LDS reads: returns what ds_read_b32 / b64 give at byte offsets 1..7
__global__ void k_lds_misaligned(uint32_t *out) {
    __shared__ uint32_t buf[16];
    if (threadIdx.x < 16) buf[threadIdx.x] = 0x03020100u + 0x04040404u * threadIdx.x; // bytes 0,1,2,...
    __syncthreads();
    uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)buf;
    if (threadIdx.x < 8) {
        uint32_t v;
        uint64_t w;
        asm volatile("ds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + threadIdx.x));
        asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(base + threadIdx.x));
        out[threadIdx.x * 3] = v;
        out[threadIdx.x * 3 + 1] = (uint32_t)w;
        out[threadIdx.x * 3 + 2] = (uint32_t)(w >> 32);
    }
}

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    int clk_khz = 0;
    CHECK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, dev));
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, clk_khz);
    const int ncu = prop.multiProcessorCount;
    uint32_t *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&out, (size_t)ncu * 4 * 1024 * 4));
    CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
#define E(n) {#n, k_##n}
    std::vector<Ent> ents = {E(and), E(xor_), E(add), E(shr_imm), E(shr_v), E(mul_u24), E(mul_hi_u24), E(mad_u24), E(mul_lo_u32),
        E(alignbit_imm), E(alignbit_v), E(alignbyte), E(perm), E(bfe), E(lshl_or), E(and_or), E(or3), E(add3), E(xad), E(lshl_add),
        E(add_sdwa_b), E(add_sdwa_bb), E(shr_sdwa), E(and_sdwa), E(mul24_sdwa), E(cndmask), E(cmp), E(addc), E(pk_mul_lo_u16),
        E(pk_add_u16), E(pk_lshrrev_b16), E(dot2_u32_u16), E(dot4_u32_u8), E(fma_f32), E(mov), E(bfi), E(mbcnt), E(ffbl), E(bpermute)};
    const int trips = 2000;
    for (int wg : {256, 1024}) { // 1 or 4 wavefronts per SIMD, one workgroup per CU
        printf("---- VALU issue: %d threads/CU (%d waves/SIMD), %d x 64 instr per wave\n", wg, wg / 256, trips);
        for (auto &e : ents) {
            hipLaunchKernelGGL(e.f, dim3(ncu), dim3(wg), 0, 0, out, 10, 1u, cyc); // warm
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.f, dim3(ncu), dim3(wg), 0, 0, out, trips, 1u, cyc);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c = 0;
            CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            const double instr_per_simd = (double)trips * 64 * (wg / 256);
            printf("%-16s %8.3f ms  wave0 clk %10llu  -> %6.2f clk/instr/SIMD (s_memtime), %6.2f ns/instr/SIMD\n", e.name, ms, c,
                   (double)c / instr_per_simd, ms * 1e6 / instr_per_simd);
        }
    }
    {
        typedef void (*lfn)(uint32_t *, int, uint32_t, unsigned long long *);
        struct L { const char *n; lfn f; } ls[] = {{"ds_read_b32 random aligned", k_lds<0>}, {"ds_read_b64 random aligned", k_lds<1>},
            {"ds_read_u8 random", k_lds<2>}, {"ds_read_u16 random", k_lds<6>}, {"ds_read_b32 random odd offsets", k_lds<3>},
            {"ds_read_b64 random 4-aligned", k_lds<4>}, {"ds_read_b32 own-bank", k_lds<5>}};
        for (auto &l : ls) CHECK(hipFuncSetAttribute((const void *)l.f, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        for (int wg : {256, 1024}) {
            printf("---- LDS: %d threads/CU, %d x 8 reads per wave\n", wg, trips);
            for (auto &l : ls) {
                hipLaunchKernelGGL(l.f, dim3(ncu), dim3(wg), 131072, 0, out, 10, 1u, cyc);
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(l.f, dim3(ncu), dim3(wg), 131072, 0, out, trips, 1u, cyc);
                CHECK(hipEventRecord(e1));
                hipError_t er = hipDeviceSynchronize();
                if (er != hipSuccess) { printf("%-32s FAILED: %s\n", l.n, hipGetErrorString(er)); continue; }
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                unsigned long long c = 0;
                CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
                const double reads_per_cu = (double)trips * 8 * (wg / 64);
                printf("%-32s %8.3f ms  wave0 clk %10llu -> %6.2f clk per wave-read per CU, %6.2f ns\n", l.n, ms, c,
                       (double)c / reads_per_cu, ms * 1e6 / reads_per_cu);
            }
        }
    }
    {
        hipLaunchKernelGGL(k_lds_misaligned, dim3(1), dim3(64), 0, 0, out);
        hipError_t er = hipDeviceSynchronize();
        if (er != hipSuccess) printf("misaligned LDS read: FAILED %s\n", hipGetErrorString(er));
        else {
            uint32_t h[24];
            CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            for (int i = 0; i < 8; i++) printf("LDS byte offset %d: b32 %08x  b64 %08x %08x\n", i, h[3 * i], h[3 * i + 1], h[3 * i + 2]);
        }
    }
    return 0;
}
