#include "encode_entry.inc"
