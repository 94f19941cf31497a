// Does the raw-buffer range check of gfx950 include the SGPR offset?  (k3_stream.h stores count rows through a buffer
// resource whose num_records covers one row segment and whose soffset selects the row.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/buffer_probe tools/buffer_probe.hip && tools/buffer_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint32_t* buf, uint32_t* res, uint32_t soff, uint32_t records)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf, 0, records, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, soff, 0);
    res[threadIdx.x] = v.x;
    const u32x4 w = {1000u + threadIdx.x, 1u, 2u, 3u};
    __builtin_amdgcn_raw_buffer_store_b128(w, r, threadIdx.x * 16, soff + 4096u, 2);
}
int main()
{
    uint32_t *buf, *res, h[4096], hr[64];
    hipMalloc(&buf, sizeof(h)); hipMalloc(&res, sizeof(hr));
    for (int i = 0; i < 4096; ++i) h[i] = 7000000u + i;
    hipMemcpy(buf, h, sizeof(h), hipMemcpyHostToDevice);
    // num_records = 256 bytes (16 lanes' worth), soffset = 1024 bytes: lanes 0..15 in range iff soffset is NOT checked
    probe<<<1, 64>>>(buf, res, 1024u, 256u);
    hipMemcpy(hr, res, sizeof(hr), hipMemcpyDeviceToHost);
    hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
    printf("load  lane 0: %u (7000256 = soffset outside the check)  lane 15: %u  lane 16: %u (0 = out of range)\n", hr[0], hr[15], hr[16]);
    printf("store lane 0 word: %u (1000 = stored)  lane 15: %u  lane 16: %u (7001344 = dropped)\n", h[(1024 + 4096) / 4], h[(1024 + 4096) / 4 + 60], h[(1024 + 4096) / 4 + 64]);
    const bool ok = hr[0] == 7000256u && hr[15] == 7000256u + 60u && hr[16] == 0u && h[(1024 + 4096) / 4] == 1000u && h[(1024 + 4096) / 4 + 64] == 7000000u + (1024 + 4096) / 4 + 64;
    printf("%s\n", ok ? "BUFFER_PROBE_OK soffset is not range-checked; lanes beyond num_records are dropped" : "BUFFER_PROBE_DIFFERENT");
    return ok ? 0 : 1;
}
