// harness.cpp -- TEST INFRASTRUCTURE ONLY.  The library's device-free host logic under AddressSanitizer + UBSan on the CPU
// box.  The product's own translation units are compiled HERE, unmodified, against the host-memory
// HIP stand-in of tests/asan/hip/hip_runtime.h; the kernel launchers they call are the no-ops below.  What runs for real:
// the GGML reader (header, vocabulary, tensor table, f16 / quantised payloads, the resident loader's block bookkeeping),
// finalize (row fusing, LayerNorm folding, conv reordering -- host loops over "device" buffers that ASan watches), the
// rnnoise-nu text parser, whisper_full's decision logic (replay_decoder / score_decoder / window_segments / the variate
// generator), the language table.  Built and driven by tests/test_host_sanitizers.py; never linked into the product.
//
//   harness load <model.bin> [resident]                      one load + set_precision(1) + free; prints the status
//   harness fuzz-ggml <model.bin> <n> <seed> <offsets.txt> [resident]
//                                                            n seeded mutations of the file IN PLACE (the test hands over a
//                                                            scratch copy), each loaded; every load must return a status
//   harness rnnoise <model.txt>                              crispy_rn_weights_from_file; prints the status
//   harness fuzz-rnnoise <model.txt> <n> <seed>
//   harness decide                                           decision-logic cases on stdin, results on stdout (see below)
#include <cinttypes>
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../crispy_amd/csrc/api_util.cpp"
#include "../../crispy_amd/csrc/asr_api.cpp"
#include "../../crispy_amd/csrc/crispy_api.cpp"
#include "../../crispy_amd/csrc/whisper_api.cpp"
#include "../../crispy_amd/csrc/ggml_load.cpp"
#include "../../crispy_amd/csrc/decode_steps.cpp"
#include "../../crispy_amd/csrc/whisper_full.cpp"

// ---- the launchers: there is no device -------------------------------------------------------------------------------
namespace crispy {
hipError_t argmax_f32(const float*, const unsigned char*, const unsigned char*, const int*, int, long, int*, int*, float*, int, hipStream_t, int, int*, int*, const StepFuse*) { return hipSuccess; }
hipError_t attn_encoder_f32(const float*, float*, int, int, int, int, hipStream_t) { return hipSuccess; }
hipError_t attn_encoder_h(const void*, const void*, void*, int, int, int, int, hipStream_t, int) { return hipSuccess; }
hipError_t attn_decoder_f32(const float*, long, const float*, long, long, long, long, long, int, const int*, float*, long, int, int, hipStream_t, AttnRows) { return hipSuccess; }
hipError_t attn_decoder_kv16(const float*, long, const void*, long, long, long, long, long, int, const int*, float*, long, int, int, hipStream_t, int, AttnRows) { return hipSuccess; }
hipError_t convert_f32_to_f16(const float*, void*, long, hipStream_t) { return hipSuccess; }
hipError_t convert_rows_f32_to_f16(const float*, long, void*, long, int, long, hipStream_t) { return hipSuccess; }
hipError_t dequant_blocks(const void*, int, long, int, void* dst, int f16, const float*, hipStream_t) { (void)dst; (void)f16; return hipSuccess; }
hipError_t embed_tokens_f32(const int*, const float*, const float*, int, const int*, float*, int, int, hipStream_t, int, const int*) { return hipSuccess; }
hipError_t embed_tokens_q(const int*, const void*, int, const float*, int, const int*, float*, int, int, hipStream_t, int, const int*) { return hipSuccess; }
bool fused_decode_supported(int, int, int) { return false; }
hipError_t fused_pack_weights(const void*, void*, int, int, hipStream_t) { return hipSuccess; }
hipError_t fused_self(const FusedSelfArgs&, bool, hipStream_t) { return hipSuccess; }
hipError_t fused_cross(const FusedCrossArgs&, hipStream_t) { return hipSuccess; }
hipError_t fused_mlp(const FusedMlpArgs&, hipStream_t) { return hipSuccess; }
hipError_t fused_finish(const FusedFinishArgs&, hipStream_t) { return hipSuccess; }
hipError_t gemm_f32_nt(const GemmArgs&, int, hipStream_t) { return hipSuccess; }
bool gemv_dec_supported(int, int) { return false; }
hipError_t gemv_dec(const GemvArgs&, int, hipStream_t) { return hipSuccess; }
hipError_t gemv_xattn(const XattnArgs&, hipStream_t) { return hipSuccess; }
bool skinny_q_supported(const GemmArgs&, int) { return false; }
hipError_t gemm_skinny_q(const GemmArgs&, hipStream_t) { return hipSuccess; }
hipError_t gemm_hh(const HGemmArgs&, int, int, hipStream_t) { return hipSuccess; }
hipError_t layernorm_f16out(const float*, const float*, const float*, void*, long, int, hipStream_t) { return hipSuccess; }
hipError_t layernorm_f32(const float*, const float*, const float*, float*, long, int, hipStream_t) { return hipSuccess; }
hipError_t mel_launch(const MelArgs&, int, hipStream_t) { return hipSuccess; }
hipError_t mel_window_launch(const MelArgs&, int, hipStream_t) { return hipSuccess; }
hipError_t pack_vocab_f16(const float*, void*, int, int, hipStream_t) { return hipSuccess; }
size_t vocab_f16_packed_bytes(int V, int K) { return (size_t)((V + 31) / 32) * 32 * K * sizeof(_Float16); }
hipError_t vocab_f16(const void*, long, const void*, float*, long, int, int, int, hipStream_t) { return hipSuccess; }
hipError_t vocab_f16_fused(const FusedIn&, const void*, float*, long, int, int, int, hipStream_t) { return hipSuccess; }
hipError_t rs_ola(const float*, float*, long, int, int, hipStream_t) { return hipSuccess; }
hipError_t rs_prep(const float*, long, long, float, int, float*, int, int, hipStream_t) { return hipSuccess; }
hipError_t rs_prep_split(const float*, long, long, float, int, void*, int, int, hipStream_t) { return hipSuccess; }
hipError_t softmax_prob_f32(const float*, int, long, int, float*, int, hipStream_t) { return hipSuccess; }
hipError_t ts_pick(const TsPickArgs&, int, hipStream_t) { return hipSuccess; }
hipError_t beam_kv_reorder(void*, void*, const int*, int, int, long, long, const int*, hipStream_t) { return hipSuccess; }
hipError_t beam_advance(const BeamArgs&, int, hipStream_t) { return hipSuccess; }
hipError_t rn_launch_frames(const RnArgs&, hipStream_t, int) { return hipSuccess; }
hipError_t rn_launch_highpass(const RnArgs&, hipStream_t, bool) { return hipSuccess; }
hipError_t rn_launch_roll_history(const RnArgs&, hipStream_t) { return hipSuccess; }
hipError_t rn_launch_tansig(const RnTables*, const float*, float*, long, int, hipStream_t) { return hipSuccess; }
}  // namespace crispy

namespace {

int load_once(const char* path, bool resident, bool verbose) {
  crispy_asr* h = nullptr;
  int rc = resident ? crispy_asr_load_resident(path, 0, &h) : crispy_asr_load(path, 0, &h);
  if (rc == CRISPY_OK && !resident) rc = crispy_asr_set_precision(h, 1);
  if (rc == CRISPY_OK) {
    // what a host does next with a loaded engine, as far as it goes without a device: vocabulary look-ups, specials,
    // language table, memory accounting
    crispy_asr_hparams hp{};
    (void)crispy_asr_hparams_get(h, &hp);
    const char* t = nullptr; size_t len = 0;
    for (int tok : {0, 1, hp.n_vocab - 1, hp.n_vocab, -1}) (void)crispy_asr_token_text(h, tok, &t, &len);
    int lt = 0;
    (void)crispy_asr_language_token(hp.n_vocab, "de", &lt);
    (void)crispy_asr_language_token(hp.n_vocab, "xx", &lt);
  }
  if (verbose) printf("{\"status\": %d, \"error\": \"%s\"}\n", rc, rc == CRISPY_OK ? "" : crispy_last_error());
  if (h) crispy_asr_free(h);
  return rc;
}

std::vector<long> read_offsets(const char* path) {
  std::vector<long> o;
  FILE* f = fopen(path, "r");
  if (!f) return o;
  long v;
  while (fscanf(f, "%ld", &v) == 1) o.push_back(v);
  fclose(f);
  return o;
}

// Mutations of a valid file, in place, undone afterwards.  Kinds: 0 truncation (four in five inside the first 4 MB --
// header, vocabulary and the small tensors -- the rest anywhere); 1 bit flips in the header (magic, hyper-parameters, filter dims); 2 a 32-bit field of a tensor header
// (n_dims, name length, type, dims) replaced by a hostile value; 3 bit flips inside a tensor header / name; 4 a vocabulary
// length field replaced.  `offs`: byte offsets of the tensor headers, then of the vocabulary length fields (negative
// separator), found by the Python side's own parser.
int fuzz_ggml(const char* path, int n, unsigned seed, const char* offsets_path, bool resident) {
  std::vector<long> offs = read_offsets(offsets_path), tens, voc;
  bool second = false;
  for (long v : offs) { if (v < 0) { second = true; continue; } (second ? voc : tens).push_back(v); }
  const int fd = open(path, O_RDWR);
  if (fd < 0) { perror("open"); return 2; }
  struct stat st{};
  if (fstat(fd, &st) != 0) return 2;
  const long size = (long)st.st_size;
  std::vector<unsigned char> orig((size_t)size);
  if (pread(fd, orig.data(), (size_t)size, 0) != size) return 2;
  std::mt19937_64 g(seed);
  auto rnd = [&](long lo, long hi) { return lo + (long)(g() % (unsigned long long)(hi - lo)); };     // [lo, hi)
  static const uint32_t hostile[] = {0u, 1u, 2u, 3u, 5u, 0x7fffffffu, 0x80000000u, 0xffffffffu, 0x10000u, 0x40000000u, 13u, 255u};
  long counts[5][3] = {};
  for (int i = 0; i < n; ++i) {
    int kind = (int)(g() % 5);
    if ((kind == 2 || kind == 3) && tens.empty()) kind = 1;
    if (kind == 4 && voc.empty()) kind = 1;
    long off = 0, len = 0;          // the byte range touched (restored afterwards); truncation: the cut
    long cut = -1;
    switch (kind) {
      case 0: cut = (g() % 5) ? rnd(0, std::min(size, 4L << 20)) : rnd(0, size); break;
      case 1: off = rnd(0, 56); len = 1; break;
      case 2: off = tens[(size_t)rnd(0, (long)tens.size())] + 4 * rnd(0, 6); len = 4; break;
      case 3: off = tens[(size_t)rnd(0, (long)tens.size())] + rnd(0, 48); len = 1; break;
      default: off = voc[(size_t)rnd(0, (long)voc.size())]; len = 4; break;
    }
    if (cut >= 0) {
      if (ftruncate(fd, cut) != 0) return 2;
    } else {
      if (off + len > size) { off = size - len; }
      unsigned char buf[4];
      memcpy(buf, orig.data() + off, (size_t)len);
      if (len == 1) buf[0] ^= (unsigned char)(1u << (g() % 8));
      else { const uint32_t v = hostile[g() % (sizeof(hostile) / sizeof(hostile[0]))]; memcpy(buf, &v, 4); }
      if (pwrite(fd, buf, (size_t)len, off) != len) return 2;
    }
    const int rc = load_once(path, resident, false);
    counts[kind][rc == CRISPY_OK ? 0 : (rc == CRISPY_ERR_BAD_MODEL || rc == CRISPY_ERR_UNSUPPORTED) ? 1 : 2]++;
    if (rc > 0 || rc < CRISPY_ERR_UNSUPPORTED) { printf("{\"fatal\": \"status %d outside crispy_status\"}\n", rc); return 3; }
    if (cut >= 0) {
      if (ftruncate(fd, size) != 0 || pwrite(fd, orig.data() + cut, (size_t)(size - cut), cut) != size - cut) return 2;
    } else if (pwrite(fd, orig.data() + off, (size_t)len, off) != len) {
      return 2;
    }
  }
  close(fd);
  printf("{\"mutations\": %d, \"by_kind\": [", n);
  for (int k = 0; k < 5; ++k) printf("%s[%ld, %ld, %ld]", k ? ", " : "", counts[k][0], counts[k][1], counts[k][2]);
  printf("], \"legend\": \"per kind (truncate, header bit, tensor field, tensor bit, vocab length): loaded, rejected as bad / unsupported model, other status\"}\n");
  return 0;
}

int rnnoise_once(const char* path, bool verbose) {
  std::vector<int8_t> w(CRISPY_RN_WEIGHT_BYTES);
  const int rc = crispy_rn_weights_from_file(path, w.data(), w.size());
  if (verbose) printf("{\"status\": %d, \"error\": \"%s\"}\n", rc, rc == CRISPY_OK ? "" : crispy_last_error());
  return rc;
}

int fuzz_rnnoise(const char* path, int n, unsigned seed) {
  FILE* f = fopen(path, "rb");
  if (!f) return 2;
  std::string orig;
  char buf[65536];
  size_t k;
  while ((k = fread(buf, 1, sizeof(buf), f)) > 0) orig.append(buf, k);
  fclose(f);
  std::mt19937_64 g(seed);
  long ok = 0, bad = 0;
  static const char* tokens[] = {"-129", "128", "99999999999999999999", "-", "x", "", "0", "1e9", "2147483648", "\n", " ", "\0"};
  for (int i = 0; i < n; ++i) {
    std::string m = orig;
    const int kind = (int)(g() % 4);
    if (kind == 0) m.resize((size_t)(g() % (orig.size() + 1)));                                   // truncation
    else if (kind == 1) for (int j = 0, nj = 1 + (int)(g() % 8); j < nj; ++j) m[(size_t)(g() % m.size())] ^= (char)(1u << (g() % 8));
    else if (kind == 2) {                                                                        // a number replaced by a hostile token
      size_t p = (size_t)(g() % m.size());
      while (p < m.size() && m[p] != ' ' && m[p] != '\n') ++p;
      size_t q = p + 1;
      while (q < m.size() && m[q] != ' ' && m[q] != '\n') ++q;
      if (p < m.size()) m.replace(p + 1, q - p - 1, tokens[g() % (sizeof(tokens) / sizeof(tokens[0]))]);
    } else {                                                                                     // a slice dropped or doubled
      const size_t p = (size_t)(g() % m.size()), l = (size_t)(g() % 4096);
      if (g() & 1) m.erase(p, l); else m.insert(p, m.substr(p, l));
    }
    f = fopen(path, "wb");
    if (!f) return 2;
    fwrite(m.data(), 1, m.size(), f);
    fclose(f);
    const int rc = rnnoise_once(path, false);
    if (rc > 0 || rc < CRISPY_ERR_UNSUPPORTED) { printf("{\"fatal\": \"status %d outside crispy_status\"}\n", rc); return 3; }
    (rc == CRISPY_OK ? ok : bad)++;
  }
  f = fopen(path, "wb");
  if (f) { fwrite(orig.data(), 1, orig.size(), f); fclose(f); }
  printf("{\"mutations\": %d, \"parsed\": %ld, \"rejected\": %ld}\n", n, ok, bad);
  return 0;
}

// decide: one case per line
//   P <n_max> <beg> <eot> <seek> <seek_end> <delta_min> <n> tok*n tid*n plog*n      -> replay + score + segments of a pass
//   U <seed> <count>                                                                   -> the first <count> variates of std::mt19937(seed)
//   N <file>                                                                           -> non_speech_token_ids of a vocabulary (one hex string per line)
int decide() {
  crispy_asr h;                           // no device state is touched: vocabulary and eot only
  char line[1 << 16];
  while (fgets(line, sizeof(line), stdin)) {
    if (line[0] == 'U') {
      unsigned seed; int count;
      if (sscanf(line + 1, "%u %d", &seed, &count) != 2) return 2;
      std::mt19937 g(seed);
      printf("[");
      for (int i = 0; i < count; ++i) printf("%s%.17g", i ? ", " : "", canonical(g));
      printf("]\n");
      continue;
    }
    if (line[0] == 'N') {             // N <file>: one vocabulary entry per line (hex) -> the ids suppress_nst masks
      char path[4096];
      if (sscanf(line + 1, "%4095s", path) != 1) return 2;
      std::vector<std::string> vocab;
      FILE* f = fopen(path, "r");
      if (!f) return 2;
      char hex[4096];
      while (fgets(hex, sizeof(hex), f)) {
        std::string t;
        for (size_t i = 0; hex[i] && hex[i + 1] && hex[i] != '\n'; i += 2) {
          unsigned v = 0;
          sscanf(hex + i, "%2x", &v);
          t.push_back((char)v);
        }
        vocab.push_back(t);
      }
      fclose(f);
      printf("[");
      bool first = true;
      for (int id : non_speech_token_ids(vocab)) { printf("%s%d", first ? "" : ", ", id); first = false; }
      printf("]\n");
      continue;
    }
    if (line[0] != 'P') continue;
    int n_max, beg, eot, seek, seek_end, delta_min, n, used = 0, adv = 0;
    if (sscanf(line + 1, "%d %d %d %d %d %d %d%n", &n_max, &beg, &eot, &seek, &seek_end, &delta_min, &n, &adv) != 7 || n < 0 || n > 4096) return 2;
    used = 1 + adv;
    std::vector<int> toks(n), tids(n);
    std::vector<float> plog(n);
    for (int i = 0; i < n; ++i) { if (sscanf(line + used, "%d%n", &toks[i], &adv) != 1) return 2; used += adv; }
    for (int i = 0; i < n; ++i) { if (sscanf(line + used, "%d%n", &tids[i], &adv) != 1) return 2; used += adv; }
    for (int i = 0; i < n; ++i) { if (sscanf(line + used, "%f%n", &plog[i], &adv) != 1) return 2; used += adv; }
    h.eot = eot;
    if ((int)h.vocab.size() != eot + 1) {
      h.vocab.resize((size_t)eot + 1);
      for (int t = 0; t <= eot; ++t) h.vocab[t] = " w" + std::to_string(t);
    }
    DecoderPass d;
    d.toks = toks.data(); d.tids = tids.data(); d.plog = plog.data(); d.n = n;
    replay_decoder(d, n_max, beg, eot, seek, seek_end, delta_min);
    if (!d.failed) score_decoder(d);
    printf("{\"failed\": %d, \"completed\": %d, \"scored\": %d, \"result_len\": %d, \"seek_delta\": %d, \"sum\": %.17g, \"avg\": %.17g, \"score\": %.17g, \"entropy\": %.17g, \"segments\": [",
           d.failed ? 1 : 0, d.completed ? 1 : 0, d.scored ? 1 : 0, d.result_len, d.seek_delta, d.sum_logprobs,
           std::isfinite(d.avg_logprobs) ? d.avg_logprobs : -1e300, std::isfinite(d.score) ? d.score : -1e300, d.entropy);
    if (!d.failed && d.result_len > 0) {
      crispy_asr_result_impl r;
      window_segments(&h, toks.data(), tids.data(), d.result_len, beg, seek, d.seek_delta, &r);
      for (size_t i = 0; i < r.seg_text.size(); ++i)
        printf("%s[%d, %d, \"%s\"]", i ? ", " : "", (int)lround(r.seg_t0[i] * 100.0), (int)lround(r.seg_t1[i] * 100.0), r.seg_text[i].c_str());
    }
    printf("]}\n");
  }
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc >= 3 && !strcmp(argv[1], "load")) return load_once(argv[2], argc > 3 && !strcmp(argv[3], "resident"), true) == CRISPY_OK ? 0 : 1;
  if (argc >= 6 && !strcmp(argv[1], "fuzz-ggml"))
    return fuzz_ggml(argv[2], atoi(argv[3]), (unsigned)atoi(argv[4]), argv[5], argc > 6 && !strcmp(argv[6], "resident"));
  if (argc >= 3 && !strcmp(argv[1], "rnnoise")) return rnnoise_once(argv[2], true) == CRISPY_OK ? 0 : 1;
  if (argc >= 5 && !strcmp(argv[1], "fuzz-rnnoise")) return fuzz_rnnoise(argv[2], atoi(argv[3]), (unsigned)atoi(argv[4]));
  if (argc >= 2 && !strcmp(argv[1], "decide")) return decide();
  fprintf(stderr, "usage: see the head of tests/asan/harness.cpp\n");
  return 2;
}
