/* placeholder, filled in below */
#include "orc_internal.h"
int orc_compress_multi(const int* k, const uint32_t* v, size_t n, size_t input_size, const uint8_t* input,
                       size_t* encoded_size, uint8_t* encoded, size_t num_threads) {
  (void)k; (void)v; (void)n; (void)input_size; (void)input; (void)encoded_size; (void)encoded; (void)num_threads;
  return 0;
}
