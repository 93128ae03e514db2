"""chimeralm_amd -- MI355X (gfx950) engine for ChimeraLM's `predict` hot path.

Host-side mirrors of the reference interface (`lm`, `basic_module`, `hyena`, `transformer`, `tokenizer`, `bam`, `callbacks`,
`__main__`) over the C ABI of `csrc/libchimeralm_hip.so` (`include/chimeralm_hip.h`, `include/chimeralm_feed.h`).  Nothing is
imported eagerly: `import chimeralm_amd` does not load the native library, `chimeralm_amd._native.load()` does and fails loudly
when it is missing.  See DESIGN.md and INTEGRATION.md."""

__version__ = "0.1.0"
