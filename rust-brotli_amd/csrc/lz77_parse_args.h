// lz77_parse_args.h -- kernel arguments of the parse launches (lz77_kernels.hip); include after lz77_chain.h
#ifndef BROTLI_MI355X_LZ77_PARSE_ARGS_H_
#define BROTLI_MI355X_LZ77_PARSE_ARGS_H_

namespace brotli_mi355x {

struct ParseArgs {
  Lz77Params P;
  ChainTables T;
  const Segment* segments;
  SegEntry* entries;
  SegExit* exits;
  uint32_t first_segment;
  const uint32_t* list;  // optional explicit segment indices
  uint8_t* sched;  // list rounds: per segment, 1 if it is in the list (see br_parse_chain)
  uint32_t count;
  uint32_t per_xcd;      // 0: identity mapping
  uint32_t max_continuation;
};

}  // namespace brotli_mi355x
#endif
