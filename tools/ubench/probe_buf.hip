// probe_buf.hip -- dev: is the scalar offset of a raw buffer load part of the range check on gfx950?  (fa_fwd_w64.hpp's LDS-DMA
// relies on the answer.)  V# with NUM_RECORDS = 64 bytes over an array a[i] = i + 1; every lane loads the dword at voffset = 4 lane
// with soffset = S.  Prints the index of the first lane that reads 0 (out of range): 16 - S/4 if soffset is checked, 16 if not.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint32_t* src, uint32_t* out, uint32_t soff, uint32_t nrec) {
    const uint64_t a = (uint64_t)src;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
    r[2] = nrec;
    r[3] = 0x00020000u;
    uint32_t v;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(threadIdx.x * 4u), "s"(r), "s"(soff) : "memory");
    out[threadIdx.x] = v;
}
int main() {
    uint32_t h[256], *d, *o, ho[64];
    for (int i = 0; i < 256; ++i) h[i] = i + 1;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 256); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (uint32_t s : {0u, 32u, 48u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, s, 64u);
        hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost);
        int first0 = 64;
        for (int i = 63; i >= 0; --i) if (ho[i] == 0) first0 = i;
        printf("soffset %2u, NUM_RECORDS 64: lane 0 reads %u, first lane reading 0: %d  -> soffset %s the range check\n", s, ho[0], first0,
               first0 == 16 ? "is NOT part of" : (first0 == 16 - (int)s / 4 ? "IS part of" : "?? (unexpected)"));
    }
    return 0;
}
