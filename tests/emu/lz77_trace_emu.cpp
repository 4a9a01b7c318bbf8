// tests/emu/lz77_trace_emu.cpp -- C entry point used by the tests to run the LZ77 stage (host driver
// + chain code) and fetch the resulting command list.  Linked against device_emu.cpp (CPU) for the
// "not gpu" tests; the product library exports the same symbol backed by the HIP kernels.
#include "../../rust-brotli_amd/csrc/lz77_trace.inc"
