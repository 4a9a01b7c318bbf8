// tests/emu: same flat entry point as the product library, linked against the CPU emulation seam
#include "../../rust-brotli_amd/csrc/encode_entry.inc"
