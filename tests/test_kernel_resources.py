"""Compiler-side regression guard (CPU: hipcc cross-compiles gfx950 without a GPU).

HISTORY.md section 4.7: what cost the hot kernels most in round 2 was not their structure but registers hipcc spilled --
scratch loads and stores are vector-memory instructions, `vmcnt` retires in order, so a reload in the middle of a tuned loop
waits for every weight set / row request issued before it.  `chimeralm_amd.build` keeps hipcc's per-kernel resource remarks of
the last build in `csrc/kernel_resources.txt`; this test holds the kernels of the headline path to (almost) no scratch."""
import re

from chimeralm_amd import build as B


def _resources():
    B.build()
    out, name = {}, None
    for ln in B.RESOURCES.read_text().splitlines():
        if ln.startswith("Function Name: "):
            name = ln.split(": ", 1)[1].strip()
            out[name] = {}
        elif name and ":" in ln:
            k, v = ln.strip().split(":", 1)
            out[name][k.strip()] = v.strip()
    return out


def _scratch(res, pattern):
    hits = {n: int(r["ScratchSize [bytes/lane]"]) for n, r in res.items() if re.search(pattern, n)}
    assert hits, f"no kernel matches {pattern}"
    return hits


def test_hot_kernels_have_no_scratch_in_their_loops():
    res = _resources()
    # bytes per lane allowed: 0 for the kernels that are > 90 % of the headline step; a few dwords elsewhere (set-up values
    # spilled outside the loops)
    budget = [
        # fp16c tail (round 4: + the lo planes of y / z / both LayerNorm tiles).  The gated in_proj variant -- 3 of 4 launches -- keeps
        # the thread index in scratch (stored in the prologue, reloaded at the top of every tile trip, before anything is requested:
        # nothing to wait behind) and, since y's lo loads moved into the in_proj hooks, two more dwords per tile -- the lane index,
        # reloaded once behind out_proj, and a staging address, reloaded once in the in_proj stage (the same-box timing of that change
        # includes them: -1.2 %); its compensated-MLP form is at zero.  The score variant spills one dword per tile
        # (stored before the y tile's barrier, reloaded behind the pooling barrier, where no weight set is awaited).
        (r"tail16_kernelILi3ELb0ELi1ELb1ELb0E", 16),
        (r"tail16_kernelILi3ELb0ELi1ELb1ELb1E", 0),
        (r"tail16_kernelILi3ELb0ELi1ELb0ELb[01]E", 0),           # (raw-rows form of the in_proj variant: A/B runs, tests)
        (r"tail16_kernelILi3ELb0ELi2E", 16),                     # fp16c tail, score variant (one more dword with the 16-column y-lo staging)
        (r"tail16_kernelILi2ELb0ELi1E", 0),                      # plain fp16
        (r"hyena_conv_pers_kernelINS_5f16_tELb0E", 0),           # 8k convolution, blocks 1-3
        (r"hyena_conv_pers_kernelINS_5f16_tELb1E", 16),          # block 0 (token ids)
        (r"hyena_conv_kernelILi13ENS_5f16_tELb0ELb0E", 0),       # 4k reads
        (r"hyena_conv_seg_kernel", 0),                            # long reads, every variant (round 5; block 0's spilled 116 B/lane before)
        (r"enc_ffn16_kernelILi[123]E", 0),                        # transformer layer kernel, all three 16-bit modes
        (r"conv3_relu_pool_kernelILi[123]E", 0),
        (r"attention_fwd_kernelILi2ELb[01]E", 0),                 # fp16 attention, one plane and hi + lo planes
        (r"tail32_kernelILi[012]ELi[01]E", 0),                    # exact fp32 / fp16x3, round 4: the fused block tail (round 5: + its score variant) ...
        (r"enc32_kernelILb[01]E", 0),                             # ... and the transformer's encoder layer, CNN stem and attention
        (r"conv32_kernel", 0),
        (r"attention32_kernel", 0),
        (r"attention_x3_kernel", 0),                              # (round 5: three waves per SIMD instead of four -- 56 B/lane of scratch gone)
    ]
    for pattern, allowed in budget:
        for name, got in _scratch(res, pattern).items():
            assert got <= allowed, f"{name}: {got} bytes of scratch per lane (budget {allowed})"


def test_tile_kernels_keep_two_waves_per_simd():
    res = _resources()
    for name, r in res.items():
        if re.search(r"tail16_kernelILi[123]ELb0", name) or "hyena_conv_pers_kernel" in name or re.search(r"(tail|enc)32_kernel", name):
            assert int(r["Occupancy [waves/SIMD]"]) >= 2, name
