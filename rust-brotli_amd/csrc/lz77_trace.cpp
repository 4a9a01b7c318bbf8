// lz77_trace.cpp -- exports brotli_mi355x_lz77_trace from the product library (HIP backed).
#include "lz77_trace.inc"
