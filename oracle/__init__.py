"""CPU oracle for the ChimeraLM `predict` hot path.  TEST INFRASTRUCTURE ONLY.

Nothing in `chimeralm_amd/` may import this package.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` use it, and
only as the checker / reported baseline -- never as the thing measured or shipped.

Parity status:
  * head, tokenizer, collator, read-name packing, prediction-file format:
    PINNED against the reference's own importable modules and known-answer
    tests (fixtures under tests/golden/, generator tests/golden/make_golden.py).
  * HyenaDNA backbone: PARITY UNPINNED.  The
    arithmetic lives in Hugging Face Hub remote code
    `LongSafari/hyenadna-small-32k-seqlen-hf` (revision unpinned by the
    reference, chimeralm/models/components/hyena.py:237) which is absent from
    /root/reference and unreachable offline; the reference holds no golden
    vector or test for it.  `hyena_oracle.py` restates the published HyenaDNA
    algorithm and anchors on the reference call sites (hyena.py:244-256).
"""
