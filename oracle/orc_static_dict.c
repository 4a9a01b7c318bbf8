/* oracle/orc_static_dict.c -- CPU restatement of BrotliFindAllStaticDictionaryMatches (quality >= 10 only).
 * TEST INFRASTRUCTURE ONLY (see brotli_oracle.h).
 *
 * Follows src/enc/static_dict.rs:
 *   Hash :37-40, IsMatch :251-288, AddMatch :290-293, DictMatchLength :296-306,
 *   BrotliFindAllStaticDictionaryMatches :309-1300 (identity / uppercase words, then the words behind a " " or "."
 *   prefix, behind "e " / "s " / ", " / U+00A0, and behind " the " / ".com/").
 * The lookup table (tables/brotli_static_dict_lut.h) is the reference's generated data (static_dict_lut.rs).
 * tools/check_static_dict.py compares the order of transform ids, length offsets and character tests of this file with
 * the reference text (build container only).
 */
#include "orc_internal.h"
#include "../tables/brotli_static_dict_lut.h"

static const uint8_t kUppercaseFirst = 10;
static const uint8_t kOmitLastNTransforms[10] = {0, 12, 27, 23, 42, 63, 56, 48, 59, 64};

typedef struct {
  uint8_t l, t;
  uint16_t i;
} DictWord;

static inline uint32_t dict_hash(const uint8_t* data) { /* :37-40, kDictHashMul32, kDictNumBits = 15 */
  uint32_t v;
  memcpy(&v, data, 4);
  return (v * 0x1e35a7bdu) >> (32 - 15);
}

static size_t match_len(const uint8_t* s1, const uint8_t* s2, size_t limit) {
  size_t i = 0;
  while (i < limit && s1[i] == s2[i]) ++i;
  return i;
}

/* :251-288 */
static int is_match(DictWord w, const uint8_t* data, size_t max_length) {
  if (w.l > max_length) return 0;
  const uint8_t* dict = orc_dictionary_data() + orc_dictionary_offsets_by_length()[w.l] + (size_t)w.l * w.i;
  if (w.t == 0) {
    return match_len(dict, data, w.l) == w.l;
  } else if (w.t == 10) {
    return dict[0] >= 'a' && dict[0] <= 'z' && (dict[0] ^ 32) == data[0] &&
           match_len(dict + 1, data + 1, (size_t)w.l - 1) == (size_t)w.l - 1;
  } else {
    for (size_t i = 0; i < w.l; ++i) {
      if (dict[i] >= 'a' && dict[i] <= 'z') {
        if ((dict[i] ^ 32) != data[i]) return 0;
      } else if (dict[i] != data[i]) {
        return 0;
      }
    }
    return 1;
  }
}

/* :290-293 */
static inline void add_match(size_t distance, size_t len, size_t len_code, uint32_t* matches) {
  uint32_t m = (uint32_t)((distance << 5) + len_code);
  if (m < matches[len]) matches[len] = m;
}

/* :296-306 */
static size_t dict_match_length(const uint8_t* data, size_t id, size_t len, size_t maxlen) {
  size_t offset = orc_dictionary_offsets_by_length()[len] + len * id;
  return match_len(orc_dictionary_data() + offset, data, ORC_MIN(len, maxlen));
}

#define ADD(t, dl) add_match(id + (size_t)(t) * n, l + (dl), l, matches)
#define NEXT_WORD()                                                      \
  uint32_t packed = kStaticDictionaryWords[offset++];                    \
  DictWord w = {(uint8_t)packed, (uint8_t)(packed >> 8), (uint16_t)(packed >> 16)}; \
  size_t l = w.l & 0x1f;                                                 \
  size_t n = (size_t)1 << orc_dictionary_size_bits_by_length()[l];       \
  size_t id = w.i;                                                       \
  end = (w.l & 0x80) != 0;                                               \
  w.l = (uint8_t)l

/* :309-1300.  matches[0..37] hold kInvalidMatch on entry. */
int orc_find_all_static_dictionary_matches(const uint8_t* data, size_t min_length, size_t max_length,
                                           uint32_t* matches) {
  int has_found_match = 0;
  {
    size_t offset = kStaticDictionaryBuckets[dict_hash(data)];
    int end = offset == 0;
    while (!end) {
      NEXT_WORD();
      if (w.t == 0) {
        size_t matchlen = dict_match_length(data, id, l, max_length);
        if (matchlen == l) {
          ADD(0, 0);
          has_found_match = 1;
        }
        if (matchlen >= l - 1) {
          add_match(id + 12 * n, l - 1, l, matches);
          if (l + 2 < max_length && data[l - 1] == 'i' && data[l] == 'n' && data[l + 1] == 'g' && data[l + 2] == ' ')
            ADD(49, 3);
          has_found_match = 1;
        }
        size_t minlen = min_length;
        if (l > 9) minlen = ORC_MAX(minlen, l - 9);
        size_t maxlen = ORC_MIN(matchlen, l - 2);
        for (size_t len = minlen; len <= maxlen; ++len) {
          add_match(id + (size_t)kOmitLastNTransforms[l - len] * n, len, l, matches);
          has_found_match = 1;
        }
        if (matchlen < l || l + 6 >= max_length) continue;
        const uint8_t* s = data + l;
        if (s[0] == ' ') {
          ADD(1, 1);
          if (s[1] == 'a') {
            if (s[2] == ' ') {
              ADD(28, 3);
            } else if (s[2] == 's') {
              if (s[3] == ' ') ADD(46, 4);
            } else if (s[2] == 't') {
              if (s[3] == ' ') ADD(60, 4);
            } else if (s[2] == 'n' && s[3] == 'd' && s[4] == ' ') {
              ADD(10, 5);
            }
          } else if (s[1] == 'b') {
            if (s[2] == 'y' && s[3] == ' ') ADD(38, 4);
          } else if (s[1] == 'i') {
            if (s[2] == 'n') {
              if (s[3] == ' ') ADD(16, 4);
            } else if (s[2] == 's' && s[3] == ' ') {
              ADD(47, 4);
            }
          } else if (s[1] == 'f') {
            if (s[2] == 'o') {
              if (s[3] == 'r' && s[4] == ' ') ADD(25, 5);
            } else if (s[2] == 'r' && s[3] == 'o' && s[4] == 'm' && s[5] == ' ') {
              ADD(37, 6);
            }
          } else if (s[1] == 'o') {
            if (s[2] == 'f') {
              if (s[3] == ' ') ADD(8, 4);
            } else if (s[2] == 'n' && s[3] == ' ') {
              ADD(45, 4);
            }
          } else if (s[1] == 'n') {
            if (s[2] == 'o' && s[3] == 't' && s[4] == ' ') ADD(80, 5);
          } else if (s[1] == 't') {
            if (s[2] == 'h') {
              if (s[3] == 'e') {
                if (s[4] == ' ') ADD(5, 5);
              } else if (s[3] == 'a' && s[4] == 't' && s[5] == ' ') {
                ADD(29, 6);
              }
            } else if (s[2] == 'o' && s[3] == ' ') {
              ADD(17, 4);
            }
          } else if (s[1] == 'w' && s[2] == 'i' && s[3] == 't' && s[4] == 'h' && s[5] == ' ') {
            ADD(35, 6);
          }
        } else if (s[0] == '"') {
          ADD(19, 1);
          if (s[1] == '>') ADD(21, 2);
        } else if (s[0] == '.') {
          ADD(20, 1);
          if (s[1] == ' ') {
            ADD(31, 2);
            if (s[2] == 'T' && s[3] == 'h') {
              if (s[4] == 'e') {
                if (s[5] == ' ') ADD(43, 6);
              } else if (s[4] == 'i' && s[5] == 's' && s[6] == ' ') {
                ADD(75, 7);
              }
            }
          }
        } else if (s[0] == ',') {
          ADD(76, 1);
          if (s[1] == ' ') ADD(14, 2);
        } else if (s[0] == '\n') {
          ADD(22, 1);
          if (s[1] == '\t') ADD(50, 2);
        } else if (s[0] == ']') {
          ADD(24, 1);
        } else if (s[0] == '\'') {
          ADD(36, 1);
        } else if (s[0] == ':') {
          ADD(51, 1);
        } else if (s[0] == '(') {
          ADD(57, 1);
        } else if (s[0] == '=') {
          if (s[1] == '"') {
            ADD(70, 2);
          } else if (s[1] == '\'') {
            ADD(86, 2);
          }
        } else if (s[0] == 'a') {
          if (s[1] == 'l' && s[2] == ' ') ADD(84, 3);
        } else if (s[0] == 'e') {
          if (s[1] == 'd') {
            if (s[2] == ' ') ADD(53, 3);
          } else if (s[1] == 'r') {
            if (s[2] == ' ') ADD(82, 3);
          } else if (s[1] == 's' && s[2] == 't' && s[3] == ' ') {
            ADD(95, 4);
          }
        } else if (s[0] == 'f') {
          if (s[1] == 'u' && s[2] == 'l' && s[3] == ' ') ADD(90, 4);
        } else if (s[0] == 'i') {
          if (s[1] == 'v') {
            if (s[2] == 'e' && s[3] == ' ') ADD(92, 4);
          } else if (s[1] == 'z' && s[2] == 'e' && s[3] == ' ') {
            ADD(100, 4);
          }
        } else if (s[0] == 'l') {
          if (s[1] == 'e') {
            if (s[2] == 's' && s[3] == 's' && s[4] == ' ') ADD(93, 5);
          } else if (s[1] == 'y' && s[2] == ' ') {
            ADD(61, 3);
          }
        } else if (s[0] == 'o' && s[1] == 'u' && s[2] == 's' && s[3] == ' ') {
          ADD(106, 4);
        }
      } else {
        int is_all_caps = w.t != kUppercaseFirst;
        if (!is_match(w, data, max_length)) continue;
        ADD(is_all_caps ? 44 : 9, 0);
        has_found_match = 1;
        if (l + 1 >= max_length) continue;
        const uint8_t* s = data + l;
        if (s[0] == ' ') {
          ADD(is_all_caps ? 68 : 4, 1);
        } else if (s[0] == '"') {
          ADD(is_all_caps ? 87 : 66, 1);
          if (s[1] == '>') ADD(is_all_caps ? 97 : 69, 2);
        } else if (s[0] == '.') {
          ADD(is_all_caps ? 101 : 79, 1);
          if (s[1] == ' ') ADD(is_all_caps ? 114 : 88, 2);
        } else if (s[0] == ',') {
          ADD(is_all_caps ? 112 : 99, 1);
          if (s[1] == ' ') ADD(is_all_caps ? 107 : 58, 2);
        } else if (s[0] == '\'') {
          ADD(is_all_caps ? 94 : 74, 1);
        } else if (s[0] == '(') {
          ADD(is_all_caps ? 113 : 78, 1);
        } else if (s[0] == '=') {
          if (s[1] == '"') {
            ADD(is_all_caps ? 105 : 104, 2);
          } else if (s[1] == '\'') {
            ADD(is_all_caps ? 116 : 108, 2);
          }
        }
      }
    }
  }
  if (max_length >= 5 && (data[0] == ' ' || data[0] == '.')) {
    int is_space = data[0] == ' ';
    size_t offset = kStaticDictionaryBuckets[dict_hash(data + 1)];
    int end = offset == 0;
    while (!end) {
      NEXT_WORD();
      if (w.t == 0) {
        if (!is_match(w, data + 1, max_length - 1)) continue;
        ADD(is_space ? 6 : 32, 1);
        has_found_match = 1;
        if (l + 2 >= max_length) continue;
        const uint8_t* s = data + l + 1;
        if (s[0] == ' ') {
          ADD(is_space ? 2 : 77, 2);
        } else if (s[0] == '(') {
          ADD(is_space ? 89 : 67, 2);
        } else if (is_space) {
          if (s[0] == ',') {
            ADD(103, 2);
            if (s[1] == ' ') ADD(33, 3);
          } else if (s[0] == '.') {
            ADD(71, 2);
            if (s[1] == ' ') ADD(52, 3);
          } else if (s[0] == '=') {
            if (s[1] == '"') {
              ADD(81, 3);
            } else if (s[1] == '\'') {
              ADD(98, 3);
            }
          }
        }
      } else if (is_space) {
        int is_all_caps = w.t != kUppercaseFirst;
        if (!is_match(w, data + 1, max_length - 1)) continue;
        ADD(is_all_caps ? 85 : 30, 1);
        has_found_match = 1;
        if (l + 2 >= max_length) continue;
        const uint8_t* s = data + l + 1;
        if (s[0] == ' ') {
          ADD(is_all_caps ? 83 : 15, 2);
        } else if (s[0] == ',') {
          if (!is_all_caps) ADD(109, 2);
          if (s[1] == ' ') ADD(is_all_caps ? 111 : 65, 3);
        } else if (s[0] == '.') {
          ADD(is_all_caps ? 115 : 96, 2);
          if (s[1] == ' ') ADD(is_all_caps ? 117 : 91, 3);
        } else if (s[0] == '=') {
          if (s[1] == '"') {
            ADD(is_all_caps ? 110 : 118, 3);
          } else if (s[1] == '\'') {
            ADD(is_all_caps ? 119 : 120, 3);
          }
        }
      }
    }
  }
  if (max_length >= 6 &&
      ((data[1] == ' ' && (data[0] == 'e' || data[0] == 's' || data[0] == ',')) || (data[0] == 0xc2 && data[1] == 0xa0))) {
    size_t offset = kStaticDictionaryBuckets[dict_hash(data + 2)];
    int end = offset == 0;
    while (!end) {
      NEXT_WORD();
      if (w.t == 0 && is_match(w, data + 2, max_length - 2)) {
        if (data[0] == 0xc2) {
          ADD(102, 2);
          has_found_match = 1;
        } else if (l + 2 < max_length && data[l + 2] == ' ') {
          size_t t = data[0] == 'e' ? 18 : (data[0] == 's' ? 7 : 13);
          ADD(t, 3);
          has_found_match = 1;
        }
      }
    }
  }
  if (max_length >= 9 && ((data[0] == ' ' && data[1] == 't' && data[2] == 'h' && data[3] == 'e' && data[4] == ' ') ||
                          (data[0] == '.' && data[1] == 'c' && data[2] == 'o' && data[3] == 'm' && data[4] == '/'))) {
    size_t offset = kStaticDictionaryBuckets[dict_hash(data + 5)];
    int end = offset == 0;
    while (!end) {
      NEXT_WORD();
      if (w.t == 0 && is_match(w, data + 5, max_length - 5)) {
        ADD(data[0] == ' ' ? 41 : 72, 5);
        has_found_match = 1;
        if (l + 5 < max_length) {
          const uint8_t* s = data + l + 5;
          if (data[0] == ' ' && l + 8 < max_length && s[0] == ' ' && s[1] == 'o' && s[2] == 'f' && s[3] == ' ') {
            ADD(62, 9);
            if (l + 12 < max_length && s[4] == 't' && s[5] == 'h' && s[6] == 'e' && s[7] == ' ') ADD(73, 13);
          }
        }
      }
    }
  }
  return has_found_match;
}
