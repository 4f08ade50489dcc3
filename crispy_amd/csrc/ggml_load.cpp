// ggml_load.cpp -- whisper.cpp GGML model files into a crispy_asr handle: header, vocabulary, tensor table, f32 / f16 / block-
// quantised payloads; `crispy_asr_load` inflates quantised matrices at load, `crispy_asr_load_resident` keeps them as the
// file's ggml blocks (managers/model.rs:99,137: whisper-medium-q4_1.bin, ggml-large-v3-q5_0.bin).  Replaces
// transcribe_rs::whisper_cpp::WhisperEngine::load (src-tauri/src/managers/transcription.rs:138-141).
#include "whisper_internal.h"

using namespace crispy;
using namespace crispy::asr;

// ---------------------------------------------------------------------------------------------
// whisper.cpp GGML model file [UPSTREAM-RECALL]:
//   u32 magic 0x67676d6c | 11 x i32 hparams (.., n_mels, ftype) | i32 n_mel, i32 n_fft, f32 filters
//   | i32 n_tokens, then (u32 len, bytes) per token | tensors until EOF:
//   i32 n_dims, i32 name_len, i32 ttype, i32 ne[n_dims] (innermost first), name, data.
// f32 / f16 tensors are taken as is; q4_0, q4_1, q5_0, q5_1, q8_0 blocks (the catalog's medium-q4_1 and
// large-v3-q5_0 files, managers/model.rs:99,137) are de-quantised to f32 at load time.
// ---------------------------------------------------------------------------------------------
namespace {

struct FileReader {
  FILE* f = nullptr;
  ~FileReader() { if (f) fclose(f); }
  bool read(void* dst, size_t n) { return fread(dst, 1, n, f) == n; }
};

float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000) << 16;
  uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else {
      exp = 127 - 15 + 1;
      while (!(man & 0x400)) { man <<= 1; --exp; }
      bits = sign | (exp << 23) | ((man & 0x3ff) << 13);
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
  else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  float out;
  std::memcpy(&out, &bits, 4);
  return out;
}


// ggml block-quantised rows -> f32 [UPSTREAM-RECALL ggml-quants]: blocks of 32 weights along the innermost
// dimension; d (and m) are f16; low nibbles are elements 0..15 of the block, high nibbles 16..31; q5 adds a
// fifth bit per element from the 32-bit mask qh.
struct QuantInfo { int block_bytes; };
bool quant_info(int ttype, QuantInfo* qi) {
  switch (ttype) {
    case 2: qi->block_bytes = 2 + 16; return true;           // q4_0
    case 3: qi->block_bytes = 2 + 2 + 16; return true;       // q4_1
    case 6: qi->block_bytes = 2 + 4 + 16; return true;       // q5_0
    case 7: qi->block_bytes = 2 + 2 + 4 + 16; return true;   // q5_1
    case 8: qi->block_bytes = 2 + 32; return true;           // q8_0
    default: return false;
  }
}
void dequant_block(int ttype, const uint8_t* b, float* y) {
  auto h = [&](const uint8_t* p) { uint16_t v; std::memcpy(&v, p, 2); return half_to_float(v); };
  if (ttype == 2) {
    const float d = h(b); const uint8_t* qs = b + 2;
    for (int j = 0; j < 16; ++j) { y[j] = ((qs[j] & 0x0F) - 8) * d; y[j + 16] = ((qs[j] >> 4) - 8) * d; }
  } else if (ttype == 3) {
    const float d = h(b), m = h(b + 2); const uint8_t* qs = b + 4;
    for (int j = 0; j < 16; ++j) { y[j] = (qs[j] & 0x0F) * d + m; y[j + 16] = (qs[j] >> 4) * d + m; }
  } else if (ttype == 6) {
    const float d = h(b); uint32_t qh; std::memcpy(&qh, b + 2, 4); const uint8_t* qs = b + 6;
    for (int j = 0; j < 16; ++j) {
      const int x0 = (qs[j] & 0x0F) | (((qh >> j) & 1) << 4);
      const int x1 = (qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
      y[j] = (x0 - 16) * d; y[j + 16] = (x1 - 16) * d;
    }
  } else if (ttype == 7) {
    const float d = h(b), m = h(b + 2); uint32_t qh; std::memcpy(&qh, b + 4, 4); const uint8_t* qs = b + 8;
    for (int j = 0; j < 16; ++j) {
      const int x0 = (qs[j] & 0x0F) | (((qh >> j) & 1) << 4);
      const int x1 = (qs[j] >> 4) | (((qh >> (j + 16)) & 1) << 4);
      y[j] = x0 * d + m; y[j + 16] = x1 * d + m;
    }
  } else {  // q8_0
    const float d = h(b); const int8_t* qs = reinterpret_cast<const int8_t*>(b + 2);
    for (int j = 0; j < 32; ++j) y[j] = qs[j] * d;
  }
}


}  // namespace

extern "C" {

namespace {
int load_impl(const char* model_path, int device, bool resident, crispy_asr** out);
}
int crispy_asr_load(const char* model_path, int device, crispy_asr** out) try {
  return load_impl(model_path, device, false, out);
} CRISPY_CATCH_RET("crispy_asr_load")

int crispy_asr_load_resident(const char* model_path, int device, crispy_asr** out) try {
  return load_impl(model_path, device, true, out);
} CRISPY_CATCH_RET("crispy_asr_load_resident")

int crispy_asr_memory_info(const crispy_asr* h, size_t* weight_bytes, size_t* quantised_bytes, size_t* scratch_bytes) try {
  if (!h) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_memory_info: NULL handle");
  size_t dense = h->derived_bytes, q = 0;
  for (const auto& kv : h->tensors)
    if (kv.second.d) dense += kv.second.n * sizeof(float);
  for (const auto& kv : h->qtensors)
    if (kv.second.owned) q += kv.second.nbytes;
  if (weight_bytes) *weight_bytes = dense + q;
  if (quantised_bytes) *quantised_bytes = q;
  if (scratch_bytes) *scratch_bytes = h->q_scratch_bytes;
  return CRISPY_OK;
} CRISPY_CATCH_RET("crispy_asr_memory_info")

namespace {
// the matrices finalize_resident consumes as ggml blocks (QRef): attention and MLP weights, the token embedding.  Any
// other 2-D tensor a file may hold quantised (whisper.cpp's own tool leaves them alone, the format does not forbid it:
// positional embeddings, the [d, 1] convolution biases) is read through T() as dense f32 and is inflated at load.
bool resident_block_name(const std::string& name) {
  if (name == "decoder.token_embedding.weight") return true;
  static const char* const tails[] = {".attn.query.weight", ".attn.key.weight", ".attn.value.weight", ".attn.out.weight",
                                      ".cross_attn.query.weight", ".cross_attn.key.weight", ".cross_attn.value.weight",
                                      ".cross_attn.out.weight", ".mlp.0.weight", ".mlp.2.weight"};
  for (const char* t : tails) {
    const size_t n = std::strlen(t);
    if (name.size() >= n && name.compare(name.size() - n, n, t) == 0) return true;
  }
  return false;
}

int load_impl(const char* model_path, int device, bool resident, crispy_asr** out) {
  if (!out) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_load: out is NULL");
  *out = nullptr;
  if (!model_path) return fail(CRISPY_ERR_INVALID_ARG, "crispy_asr_load: NULL path");
  FileReader r;
  r.f = fopen(model_path, "rb");
  if (!r.f) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: cannot open '%s'", model_path);
  uint32_t magic = 0;
  int32_t hpv[11];
  if (!r.read(&magic, 4) || magic != 0x67676d6c)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: '%s' is not a ggml whisper model (bad magic)", model_path);
  if (!r.read(hpv, sizeof(hpv))) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated header");
  crispy_asr_hparams hp;
  hp.n_vocab = hpv[0]; hp.n_audio_ctx = hpv[1]; hp.n_audio_state = hpv[2]; hp.n_audio_head = hpv[3];
  hp.n_audio_layer = hpv[4]; hp.n_text_ctx = hpv[5]; hp.n_text_state = hpv[6]; hp.n_text_head = hpv[7];
  hp.n_text_layer = hpv[8]; hp.n_mels = hpv[9];
  int32_t fm = 0, ff = 0;
  if (!r.read(&fm, 4) || !r.read(&ff, 4) || fm != hp.n_mels || ff != MEL_BINS)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: mel filter block is %d x %d, expected %d x %d", fm, ff, hp.n_mels,
                MEL_BINS);
  std::vector<float> filters((size_t)fm * ff);
  if (!r.read(filters.data(), filters.size() * 4)) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated filters");
  if (hp.n_vocab <= 0 || hp.n_vocab > 65536)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: n_vocab %d is not a whisper vocabulary size", hp.n_vocab);
  int32_t n_tok = 0;
  if (!r.read(&n_tok, 4) || n_tok < 0 || n_tok > hp.n_vocab + 1024)
    return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: bad vocabulary size %d", n_tok);
  std::vector<std::string> vocab(n_tok);
  for (int i = 0; i < n_tok; ++i) {
    uint32_t len = 0;
    if (!r.read(&len, 4) || len > 4096) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: bad token %d", i);
    vocab[i].resize(len);
    if (len && !r.read(&vocab[i][0], len)) return fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated vocabulary");
  }
  crispy_asr* h = nullptr;
  int rc = crispy_asr_create(&hp, filters.data(), device, &h);
  if (rc != CRISPY_OK) return rc;
  h->vocab = std::move(vocab);
  // (h->resident is decided after the tensor loop: only a file that HAS quantised matrices takes the resident path)
  auto bail = [&](int code) {
    const std::string keep = last_error_cstr();
    crispy_asr_free(h);
    return fail(code, "%s", keep.c_str());
  };
  std::vector<float> buf;
  std::vector<uint16_t> hbuf;
  std::vector<uint8_t> qbuf;
  // a tensor is read only if the model needs it and the file's shape has exactly the element count the
  // hyper-parameters imply: the buffers below are sized from this table, never from numbers a corrupt file supplies
  const std::map<std::string, size_t> expect = expected_tensors(hp);
  for (;;) {
    int32_t n_dims = 0, name_len = 0, ttype = 0;
    if (!r.read(&n_dims, 4)) break;  // clean EOF
    if (!r.read(&name_len, 4) || !r.read(&ttype, 4) || n_dims < 1 || n_dims > 4 || name_len <= 0 || name_len > 256) {
      fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: corrupt tensor header");
      return bail(CRISPY_ERR_BAD_MODEL);
    }
    int32_t ne[4] = {1, 1, 1, 1};
    unsigned long long n64 = 1;          // <= (2^31)^4 would overflow: checked against 2^40 after every factor
    for (int i = 0; i < n_dims; ++i) {
      if (!r.read(&ne[i], 4) || ne[i] <= 0) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: corrupt tensor shape"); return bail(CRISPY_ERR_BAD_MODEL); }
      n64 *= (unsigned long long)ne[i];
      if (n64 > (1ull << 40)) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: corrupt tensor shape (element count overflows)"); return bail(CRISPY_ERR_BAD_MODEL); }
    }
    const size_t n = (size_t)n64;
    std::string name(name_len, '\0');
    if (!r.read(&name[0], name_len)) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated tensor name"); return bail(CRISPY_ERR_BAD_MODEL); }
    {
      const auto it = expect.find(name);
      if (it == expect.end()) {
        fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: unknown tensor '%s' in model file", name.c_str());
        return bail(CRISPY_ERR_BAD_MODEL);
      }
      if (it->second != n) {
        fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: tensor '%s' has %zu elements, the hyper-parameters imply %zu",
             name.c_str(), n, it->second);
        return bail(CRISPY_ERR_BAD_MODEL);
      }
    }
    buf.resize(n);
    if (ttype == 0) {
      if (!r.read(buf.data(), n * 4)) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated data of '%s'", name.c_str()); return bail(CRISPY_ERR_BAD_MODEL); }
    } else if (ttype == 1) {
      hbuf.resize(n);
      if (!r.read(hbuf.data(), n * 2)) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated data of '%s'", name.c_str()); return bail(CRISPY_ERR_BAD_MODEL); }
      for (size_t i = 0; i < n; ++i) buf[i] = half_to_float(hbuf[i]);
    } else {
      QuantInfo qi;
      if (!quant_info(ttype, &qi) || ne[0] % 32 != 0) {
        fail(CRISPY_ERR_UNSUPPORTED, "crispy_asr_load: tensor '%s' has ggml type %d (supported: f32 0, f16 1, q4_0 2, "
             "q4_1 3, q5_0 6, q5_1 7, q8_0 8; rows must be multiples of 32)", name.c_str(), ttype);
        return bail(CRISPY_ERR_UNSUPPORTED);
      }
      const size_t n_blocks = n / 32;
      qbuf.resize(n_blocks * qi.block_bytes);
      if (!r.read(qbuf.data(), qbuf.size())) { fail(CRISPY_ERR_BAD_MODEL, "crispy_asr_load: truncated data of '%s'", name.c_str()); return bail(CRISPY_ERR_BAD_MODEL); }
      if (resident && n_dims == 2 && resident_block_name(name)) {
        // the blocks stay as they are (managers/model.rs:99,137: the catalog's q4_1 / q5_0 files): no f32 tensor is made
        QTensor q;
        q.ttype = ttype; q.n = n; q.cols = ne[0]; q.nbytes = qbuf.size();
        if (hipSetDevice(device) != hipSuccess || hipMalloc(&q.d, q.nbytes + 16) != hipSuccess ||      // (+16: the in-register block fetch reads whole dwords)
            hipMemcpy(q.d, qbuf.data(), q.nbytes, hipMemcpyHostToDevice) != hipSuccess) {
          if (q.d) (void)hipFree(q.d);
          fail(CRISPY_ERR_OOM, "crispy_asr_load_resident: no device memory for '%s' (%zu bytes)", name.c_str(), q.nbytes);
          return bail(CRISPY_ERR_OOM);
        }
        h->qtensors[name] = q;
        h->tensors[name].set = true;
        continue;
      }
      for (size_t bi = 0; bi < n_blocks; ++bi) dequant_block(ttype, qbuf.data() + bi * qi.block_bytes, buf.data() + bi * 32);
    }
    rc = crispy_asr_set_tensor(h, name.c_str(), buf.data(), n);
    if (rc != CRISPY_OK) return bail(rc);
  }
  // A file without a single quantised matrix (f32 / f16: the catalog's ggml-small.bin and large-v3-turbo,
  // managers/model.rs:80,118) loads exactly as crispy_asr_load does: dense tensors, the ordinary finalize, the f16 copies
  // of precision mode 1 -- not the resident path, where every matrix would be copied into the scratch slot in front of
  // every product.
  h->resident = resident && !h->qtensors.empty();
  rc = crispy_asr_finalize(h);
  if (rc != CRISPY_OK) return bail(rc);
  if (resident && !h->resident) {              // what crispy_asr_load_resident promises: whisper.cpp's arithmetic
    rc = crispy_asr_set_precision(h, 1);
    if (rc != CRISPY_OK) return bail(rc);
  }
  *out = h;
  return CRISPY_OK;
}
}  // namespace

}  // extern "C"
