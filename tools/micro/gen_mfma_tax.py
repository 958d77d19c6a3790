#!/usr/bin/env python
"""Generate tools/micro/mfma_tax.hip: what does ONE extra instruction cost an fp32-MFMA stream on gfx950?

Every kernel is the same hand-written loop -- per step 12 x v_mfma_f32_16x16x4_f32 on three accumulators (the
per-frequency inner loop of csrc/conv_wino.hip: 3 co sub-tiles x 4 k-steps, operand fragments double-buffered
in registers) -- plus a variable number of companions placed between the MFMAs:
   ds_read_b128 / _b64 / _b32 (conflict-free), buffer_load_dwordx4 (L2 resident), LDS-DMA pieces,
   plain VALU adds, s_barrier, and the accumulators / B operands in AGPRs instead of VGPRs.
The bodies are inline asm with fixed registers, so the instruction stream is exactly what is written here
(hipcc only wraps the loop).  Output of the binary: cycles per MFMA of every variant at 1 and 2 waves per
SIMD; the difference to the bare stream divided by the number of companions is the "tax" per instruction.

    python tools/micro/gen_mfma_tax.py > tools/micro/mfma_tax.hip
    hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_tax tools/micro/mfma_tax.hip
"""

# fixed registers
ACC_V = [0, 4, 8]           # 3 accumulators x 4 regs (v or a)
AF = 12                     # v12..15: A operands of the 4 k-steps
BF = [16, 28]               # two sets of 3 x 4 B operand regs: v16..27, v28..39 (or a16.. / a28..)
XR = 40                     # v40..63: destinations of reads beyond the B fragments
ADDR = 64                   # v64: LDS byte address (lane * 16), v65: lane * 8, v66: lane*4
GOFF = 67                   # v67: global byte offset lane*16
VAL = 68                    # v68..99 VALU scratch


def body(step, cfg):
    """One step: consumes B set `step & 1`, fills the other."""
    cur, nxt = BF[step & 1], BF[(step + 1) & 1]
    acc = 'a' if cfg.get('acc_agpr') else 'v'
    bcls = 'a' if cfg.get('b_agpr') else 'v'
    out = []
    # the companions of this step, in issue order
    comp = []
    r = 0
    for i in range(cfg.get('r128', 0)):
        dst = nxt + 4 * i if i < 3 else XR + 4 * (i - 3)
        cls = bcls if i < 3 else 'v'
        comp.append('ds_read_b128 %s[%d:%d], v%d offset:%d' % (cls, dst, dst + 3, ADDR, 1024 * (i + 4 * (step & 1))))
    for i in range(cfg.get('r64', 0)):
        dst = nxt + 2 * i if i < 6 else XR + 2 * (i - 6)
        comp.append('ds_read_b64 v[%d:%d], v%d offset:%d' % (dst, dst + 1, ADDR + 1, 512 * (i + 8 * (step & 1))))
    for i in range(cfg.get('r32', 0)):
        dst = nxt + i if i < 12 else XR + (i - 12)
        comp.append('ds_read_b32 v%d, v%d offset:%d' % (dst, ADDR + 2, 256 * (i + 16 * (step & 1))))
    for i in range(cfg.get('gl', 0)):
        dst = nxt + 4 * i if i < 3 else XR + 4 * (i - 3)
        comp.append('buffer_load_dwordx4 v[%d:%d], v%d, s[24:27], 0 offen offset:%d' % (dst, dst + 3, GOFF, 1024 * i))
    for i in range(cfg.get('dma', 0)):
        comp.append('s_mov_b32 m0, s%d\n\ts_nop 0\n\tbuffer_load_dwordx4 v%d, s[24:27], 0 offen offset:%d lds'
                    % (28 + (i & 1), GOFF, 1024 * i))
    for i in range(cfg.get('valu', 0)):
        d = VAL + (i % 16)
        comp.append('v_add_f32 v%d, v%d, v%d' % (d, VAL + 16 + (i % 16), d))
    # wait for what the previous step issued into `cur`
    waits = []
    if cfg.get('r128', 0) or cfg.get('r64', 0) or cfg.get('r32', 0):
        waits.append('lgkmcnt(0)')
    if cfg.get('gl', 0) or cfg.get('dma', 0):
        waits.append('vmcnt(0)')
    if waits and not cfg.get('late_wait'):
        out.append('s_waitcnt ' + ' '.join(waits))
    if cfg.get('barrier'):
        out.append('s_barrier')
    mf = []
    for s in range(4):
        for n in range(3):
            a = ACC_V[n]
            mf.append('v_mfma_f32_16x16x4_f32 %s[%d:%d], v%d, %s%d, %s[%d:%d]'
                      % (acc, a, a + 3, AF + s, bcls, cur + 4 * n + s, acc, a, a + 3))
    # spread the companions between the MFMAs (from the second MFMA on)
    slots = [[] for _ in range(12)]
    if cfg.get('burst'):
        slots[0] = comp
    else:
        for i, c in enumerate(comp):
            slots[min(11, (i * 11) // max(len(comp), 1))].append(c)
    for i in range(12):
        out.append(mf[i])
        out.extend(slots[i])
    return out


VARIANTS = [
    ('bare', {}),
    ('r128x1', dict(r128=1)), ('r128x2', dict(r128=2)), ('r128x3', dict(r128=3)), ('r128x4', dict(r128=4)),
    ('r128x6', dict(r128=6)), ('r128x9', dict(r128=9)),
    ('r128x3 burst', dict(r128=3, burst=1)),
    ('r64x6', dict(r64=6)), ('r64x12', dict(r64=12)),
    ('r32x12', dict(r32=12)),
    ('acc in AGPR, bare', dict(acc_agpr=1)), ('acc in AGPR, r128x3', dict(acc_agpr=1, r128=3)),
    ('acc in AGPR, r128x6', dict(acc_agpr=1, r128=6)),
    ('B in AGPR, r128x3', dict(b_agpr=1, r128=3)), ('acc+B in AGPR, r128x3', dict(acc_agpr=1, b_agpr=1, r128=3)),
    ('valu x8', dict(valu=8)), ('valu x16', dict(valu=16)), ('valu x32', dict(valu=32)), ('valu x64', dict(valu=64)),
    ('valu x16 + r128x3', dict(valu=16, r128=3)), ('valu x32 + r128x3', dict(valu=32, r128=3)),
    ('acc AGPR, valu x32 + r128x3', dict(acc_agpr=1, valu=32, r128=3)),
    ('gl x1', dict(gl=1)), ('gl x3', dict(gl=3)),
    ('dma x1', dict(dma=1)), ('dma x2', dict(dma=2)),
    ('barrier, r128x3', dict(barrier=1, r128=3)), ('barrier, bare', dict(barrier=1)),
    ('wino8-like: r128x5 valu x8 dma x1', dict(r128=5, valu=8, dma=1)),
    ('wino8-like, acc AGPR', dict(r128=5, valu=8, dma=1, acc_agpr=1)),
]


def kernel(idx, cfg):
    lines = []
    for step in range(2):
        lines += body(step, cfg)
    asm = '\\n\\t'.join(l.replace('\n\t', '\\n\\t') for l in lines)
    # defined operand values (register garbage would make the variants' data -- and clocks -- differ)
    init = '\\n\\t'.join(['v_mov_b32 v%d, 0.5' % i for i in range(0, 64)] + ['v_mov_b32 v%d, 1.0' % i for i in range(68, 100)]
                           + ['v_accvgpr_write_b32 a%d, v0' % i for i in range(40)])
    clob = ['"v%d"' % i for i in range(100)] + ['"a%d"' % i for i in range(40)] + \
        ['"s%d"' % i for i in (20, 24, 25, 26, 27, 28, 29)] + ['"scc"', '"m0"', '"memory"']
    return '''
__global__ __launch_bounds__(512, 2) void k%d(float* out, const float* g, int iters) {
  extern __shared__ float4 sm[];
  for (int i = threadIdx.x; i < 6144; i += blockDim.x) sm[i] = make_float4(1e-3f * (i & 255), 1.f, 0.5f, 0.25f);
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long ga = (unsigned long long)g;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(65536u + wv * 1024u);
  const unsigned lds1 = __builtin_amdgcn_readfirstlane(81920u + wv * 1024u);
  asm volatile(
      "s_mov_b32 s24, %%4\\n\\ts_mov_b32 s25, %%5\\n\\ts_mov_b32 s26, 0x10000\\n\\ts_mov_b32 s27, 0x00020000\\n\\t"
      "s_mov_b32 s28, %%6\\n\\ts_mov_b32 s29, %%7\\n\\t"
      "v_mov_b32 v64, %%1\\n\\tv_mov_b32 v65, %%2\\n\\tv_mov_b32 v66, %%3\\n\\tv_mov_b32 v67, %%1\\n\\t"
      "s_mov_b32 s20, %%0\\n\\t"
      "%s\\n\\t"
      "L%d_%%=:\\n\\t"
      "%s\\n\\t"
      "s_sub_u32 s20, s20, 1\\n\\ts_cmp_lg_u32 s20, 0\\n\\ts_cbranch_scc1 L%d_%%=\\n\\t"
      "s_waitcnt vmcnt(0) lgkmcnt(0)"
      :: "s"(iters), "v"(lane * 16u), "v"(lane * 8u), "v"(lane * 4u), "s"((unsigned)ga),
         "s"((unsigned)(ga >> 32) & 0xffffu), "s"(lds0), "s"(lds1)
      : %s);
  if (iters < 0) out[0] = 1.f;
}
''' % (idx, init, idx, asm, idx, ', '.join(clob))


def main():
    print('// GENERATED by tools/micro/gen_mfma_tax.py -- do not edit')
    print('#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstring>')
    for i, (name, cfg) in enumerate(VARIANTS):
        print(kernel(i, cfg))
    print('typedef void (*kfn)(float*, const float*, int);')
    print('static const struct { const char* name; kfn f; int comps; } K[] = {')
    for i, (name, cfg) in enumerate(VARIANTS):
        comps = sum(cfg.get(k, 0) for k in ('r128', 'r64', 'r32', 'gl', 'dma', 'valu'))
        print('  {"%s", k%d, %d},' % (name, i, comps))
    print('};')
    print(r'''
int main(int argc, char** argv) {
  float *out, *g;
  hipMalloc(&out, 64);
  hipMalloc(&g, 1 << 20);
  hipMemset(g, 0, 1 << 20);
  const int cus = 256, iters = 2000;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("# %s, %d CUs; per step: 12 x v_mfma_f32_16x16x4_f32 (384 pipe cycles) + companions; cycles at 2.4 GHz nominal\n",
         prop.name, prop.multiProcessorCount);
  double bare[3] = {0, 0, 0};
  for (int th : {256, 512}) {
    for (unsigned i = 0; i < sizeof(K) / sizeof(K[0]); ++i) {
      hipFuncSetAttribute((const void*)K[i].f, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipLaunchKernelGGL(K[i].f, dim3(cus), dim3(th), 131072, 0, out, g, 10);
      hipDeviceSynchronize();
      float best = 1e30f;
      for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(K[i].f, dim3(cus), dim3(th), 131072, 0, out, g, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const int wps = th / 256;
      const double steps = 2.0 * iters;                       // per wave
      const double cyc_per_step_simd = best * 1e-3 * 2.4e9 / steps;   // all waves of a SIMD together
      const double per_mfma = cyc_per_step_simd / (12.0 * wps);
      const double tf = (double)cus * (th / 64) * steps * 12 * 2048.0 / (best * 1e-3) / 1e12;
      if (i == 0) bare[wps] = cyc_per_step_simd;
      const double tax = K[i].comps ? (cyc_per_step_simd - bare[wps]) / (K[i].comps * wps) : 0.0;
      printf("%-40s waves/SIMD %d  %7.3f ms  %6.1f TFLOP/s  %5.1f cyc/MFMA  tax/companion %6.1f cyc\n", K[i].name, wps,
             best, tf, per_mfma, tax);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
  return 0;
}''')


if __name__ == '__main__':
    main()
