// synth_gen.c -- fast generators for the synthetic workloads of SURVEY.md 8(d), bit-identical to tests/synth.py
// (tests/test_synth.py checks that).  Bench / test infrastructure: the Python generators run at ~8 MB/s, too slow for the
// BASELINE sizes (256 MiB ... 4 GiB); these run at several hundred MB/s.  Not part of the product library.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint64_t xs_next(uint64_t* x) {
  uint64_t v = *x;
  v ^= v >> 12;
  v ^= v << 25;
  v ^= v >> 27;
  *x = v;
  return v * 0x2545F4914F6CDD1Dull;
}
static uint64_t xs_seed(uint64_t seed) { return seed ? seed : 0x9E3779B97F4A7C15ull; }

// synth.random_bytes: the xorshift64* stream, 8 little-endian bytes per draw
void synth_xorshift(uint64_t seed, size_t nbytes, uint8_t* out) {
  uint64_t x = xs_seed(seed);
  size_t i = 0;
  for (; i + 8 <= nbytes; i += 8) {
    const uint64_t v = xs_next(&x);
    memcpy(out + i, &v, 8);
  }
  if (i < nbytes) {
    const uint64_t v = xs_next(&x);
    memcpy(out + i, &v, nbytes - i);
  }
}

// ---- word-bigram Markov chain over the tokens of a corpus (synth.markov_text) ----
typedef struct {
  const uint8_t* corpus;
  uint32_t ntoks;
  uint32_t* tok_off;   // per token occurrence: offset / length in the corpus
  uint32_t* tok_len;
  uint32_t* tok_id;    // canonical id (index of the first occurrence with the same bytes)
  uint32_t* fol_first; // per canonical id: range in fol[] of the followers, in order of appearance
  uint32_t* fol_count;
  uint32_t* fol;       // canonical ids
} Markov;

static int is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }

static uint64_t hash_bytes(const uint8_t* p, uint32_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint32_t i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
  return h;
}

void* synth_markov_new(const uint8_t* corpus, size_t len) {
  Markov* m = (Markov*)calloc(1, sizeof(Markov));
  uint8_t* copy = (uint8_t*)malloc(len + 1);
  memcpy(copy, corpus, len);
  m->corpus = copy;
  uint32_t cap = (uint32_t)(len / 2 + 2);
  m->tok_off = (uint32_t*)malloc(cap * 4);
  m->tok_len = (uint32_t*)malloc(cap * 4);
  uint32_t n = 0;
  size_t i = 0;
  while (i < len) {
    while (i < len && is_space(copy[i])) ++i;
    if (i >= len) break;
    size_t j = i;
    while (j < len && !is_space(copy[j])) ++j;
    m->tok_off[n] = (uint32_t)i;
    m->tok_len[n] = (uint32_t)(j - i);
    ++n;
    i = j;
  }
  m->ntoks = n;
  m->tok_id = (uint32_t*)malloc((n + 1) * 4);
  // open-addressing table: slot -> token occurrence index of the canonical token
  uint32_t tsize = 1;
  while (tsize < 2 * n + 16) tsize <<= 1;
  uint32_t* table = (uint32_t*)malloc((size_t)tsize * 4);
  memset(table, 0xff, (size_t)tsize * 4);
  for (uint32_t t = 0; t < n; ++t) {
    uint64_t h = hash_bytes(copy + m->tok_off[t], m->tok_len[t]);
    uint32_t s = (uint32_t)h & (tsize - 1);
    for (;;) {
      const uint32_t o = table[s];
      if (o == 0xffffffffu) {
        table[s] = t;
        m->tok_id[t] = t;
        break;
      }
      if (m->tok_len[o] == m->tok_len[t] && memcmp(copy + m->tok_off[o], copy + m->tok_off[t], m->tok_len[t]) == 0) {
        m->tok_id[t] = o;
        break;
      }
      s = (s + 1) & (tsize - 1);
    }
  }
  free(table);
  m->fol_first = (uint32_t*)calloc(n + 1, 4);
  m->fol_count = (uint32_t*)calloc(n + 1, 4);
  m->fol = (uint32_t*)malloc((size_t)(n + 1) * 4);
  for (uint32_t t = 0; t + 1 < n; ++t) m->fol_count[m->tok_id[t]]++;
  uint32_t at = 0;
  for (uint32_t t = 0; t < n; ++t) {
    m->fol_first[t] = at;
    at += m->fol_count[t];
    m->fol_count[t] = 0;
  }
  for (uint32_t t = 0; t + 1 < n; ++t) {
    const uint32_t a = m->tok_id[t];
    m->fol[m->fol_first[a] + m->fol_count[a]++] = m->tok_id[t + 1];
  }
  return m;
}

void synth_markov_free(void* h) {
  Markov* m = (Markov*)h;
  if (!m) return;
  free((void*)m->corpus);
  free(m->tok_off);
  free(m->tok_len);
  free(m->tok_id);
  free(m->fol_first);
  free(m->fol_count);
  free(m->fol);
  free(m);
}

void synth_markov_text(void* h, uint64_t seed, size_t nbytes, uint8_t* out) {
  const Markov* m = (const Markov*)h;
  uint64_t x = xs_seed(seed);
  size_t n = 0;
  uint32_t cur = m->tok_id[0];
  uint32_t col = 0;
  while (n < nbytes) {
    const uint32_t len = m->tok_len[cur];
    const size_t take = len < nbytes - n ? len : nbytes - n;
    memcpy(out + n, m->corpus + m->tok_off[cur], take);
    n += take;
    if (n >= nbytes) break;
    col += len;
    if (col >= 70) {
      out[n++] = '\n';
      col = 0;
    } else {
      out[n++] = ' ';
      col += 1;
    }
    const uint32_t nc = m->fol_count[cur];
    if (nc == 0) {
      cur = m->tok_id[xs_next(&x) % m->ntoks];
    } else {
      cur = m->fol[m->fol_first[cur] + xs_next(&x) % nc];
    }
  }
}
