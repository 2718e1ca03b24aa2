// cjs_napi.cc — thin N-API shim over the C ABI of libcjs_hip.so (include/cjs_hip.h).
// It does no compression work: it dlopen()s the HIP library that sits next to the package, hands it
// the bytes of a Uint8Array/Buffer and wraps the malloc'd result as an external ArrayBuffer whose
// finalizer calls cjs_free.  Coercion of streams/arrays and error -> exception mapping live in the JS
// fronts (Bzip2.js / BWTC.js), exactly where the reference does them (Util.coerceInputStream etc.).
#include <node_api.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "cjs_hip.h"

namespace {

struct Api {
  void* handle = nullptr;
  int (*bzip2_compress)(const uint8_t*, size_t, int, uint8_t**, size_t*, const cjs_opts*) = nullptr;
  int (*bzip2_decompress)(const uint8_t*, size_t, int, uint8_t**, size_t*, const cjs_opts*) = nullptr;
  int (*bwtc_compress)(const uint8_t*, size_t, int, uint8_t**, size_t*, const cjs_opts*) = nullptr;
  int (*bwtc_decompress)(const uint8_t*, size_t, uint8_t**, size_t*, const cjs_opts*) = nullptr;
  long (*bzip2_table)(const uint8_t*, size_t, int, uint64_t*, uint32_t*, long, const cjs_opts*) = nullptr;
  int (*bzip2_decompress_block)(const uint8_t*, size_t, uint64_t, uint8_t**, size_t*, const cjs_opts*) = nullptr;
  void (*free_)(void*) = nullptr;
  const char* (*strerror_)(int) = nullptr;
  const char* (*detail_)(void) = nullptr;
  int (*device_count)(void) = nullptr;
  const char* (*version)(void) = nullptr;
  void (*trim)(void) = nullptr;
  std::string error;
} api;

bool load_api() {
  if (api.handle) return true;
  Dl_info info;
  std::string dir = ".";
  if (dladdr((void*)&load_api, &info) && info.dli_fname) {
    dir = info.dli_fname;
    size_t p = dir.find_last_of('/');
    dir = p == std::string::npos ? "." : dir.substr(0, p);
  }
  const char* env = getenv("CJS_HIP_LIB");
  std::string path = env ? env : dir + "/../libcjs_hip.so";
  api.handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!api.handle) { api.error = std::string("cannot load libcjs_hip.so: ") + dlerror(); return false; }
#define SYM(field, name) *(void**)(&api.field) = dlsym(api.handle, name); if (!api.field) { api.error = "missing symbol " name; api.handle = nullptr; return false; }
  SYM(bzip2_compress, "cjs_bzip2_compress") SYM(bzip2_decompress, "cjs_bzip2_decompress")
  SYM(bwtc_compress, "cjs_bwtc_compress") SYM(bwtc_decompress, "cjs_bwtc_decompress")
  SYM(bzip2_table, "cjs_bzip2_table") SYM(bzip2_decompress_block, "cjs_bzip2_decompress_block")
  SYM(free_, "cjs_free") SYM(strerror_, "cjs_strerror") SYM(detail_, "cjs_last_error_detail") SYM(device_count, "cjs_device_count") SYM(version, "cjs_version") SYM(trim, "cjs_trim")
#undef SYM
  return true;
}

// Results are handed to JS as EXTERNAL ArrayBuffers over the library's (pinned, pooled) result buffers; V8 is told how much
// memory hangs on each one, so that it collects dropped results soon and their buffers go back to the pool instead of a new
// pinned buffer being made for every call (hipHostMalloc of a 32 MB result costs more than compressing 50 MB).
void finalize_buf(napi_env env, void* data, void* hint) {
  if (api.free_) api.free_(data);
  int64_t adj = 0;
  napi_adjust_external_memory(env, -(int64_t)(intptr_t)hint, &adj);
}

napi_value throw_code(napi_env env, int code) {
  napi_value err, msg, num;
  const char* text = api.strerror_ ? api.strerror_(code) : "cjs error";
  napi_create_string_utf8(env, text, NAPI_AUTO_LENGTH, &msg);
  napi_create_error(env, nullptr, msg, &err);
  napi_create_int32(env, code, &num);
  napi_set_named_property(env, err, "cjsCode", num);
  const char* detail = api.detail_ ? api.detail_() : "";        // the reference's optDetail (J/Bzip2_joined_.js:1385-1391)
  if (detail && detail[0]) {
    napi_value d;
    napi_create_string_utf8(env, detail, NAPI_AUTO_LENGTH, &d);
    napi_set_named_property(env, err, "cjsDetail", d);
  }
  napi_throw(env, err);
  return nullptr;
}

bool get_bytes(napi_env env, napi_value v, const uint8_t** p, size_t* n) {
  bool is_ta = false, is_buf = false;
  napi_is_buffer(env, v, &is_buf);
  if (is_buf) { void* d; napi_get_buffer_info(env, v, &d, n); *p = (const uint8_t*)d; return true; }
  napi_is_typedarray(env, v, &is_ta);
  if (is_ta) {
    napi_typedarray_type t; size_t len; void* d; napi_value ab; size_t off;
    napi_get_typedarray_info(env, v, &t, &len, &d, &ab, &off);
    if (t != napi_uint8_array && t != napi_uint8_clamped_array && t != napi_int8_array) return false;
    *p = (const uint8_t*)d; *n = len; return true;
  }
  return false;
}

napi_value wrap_result(napi_env env, uint8_t* data, size_t n) {
  napi_value ab, ta;
  int64_t adj = 0;
  if (napi_create_external_arraybuffer(env, data, n, finalize_buf, (void*)(intptr_t)n, &ab) == napi_ok) napi_adjust_external_memory(env, (int64_t)n, &adj);
  else {
    // some runtimes forbid external buffers: copy instead
    void* dst;
    napi_create_arraybuffer(env, n, &dst, &ab);
    memcpy(dst, data, n);
    api.free_(data);
  }
  napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &ta);
  return ta;
}

template <int KIND>   // 0 bzip2 compress, 1 bzip2 decompress, 2 bwtc compress, 3 bwtc decompress
napi_value call_stream(napi_env env, napi_callback_info info) {
  if (!load_api()) { napi_throw_error(env, nullptr, api.error.c_str()); return nullptr; }
  size_t argc = 3; napi_value argv[3];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  const uint8_t* p = nullptr; size_t n = 0;
  if (argc < 1 || !get_bytes(env, argv[0], &p, &n)) { napi_throw_type_error(env, nullptr, "expected a Uint8Array or Buffer"); return nullptr; }
  int32_t arg = KIND == 1 ? 0 : 9;
  if (argc >= 2) napi_get_value_int32(env, argv[1], &arg);
  uint32_t flags = 0;                                           // third argument: cjs_opts.flags (bwtcCompress: CJS_FLAG_SIZE_UNKNOWN)
  if (argc >= 3) napi_get_value_uint32(env, argv[2], &flags);
  cjs_opts opts; memset(&opts, 0, sizeof opts);
  opts.struct_size = sizeof opts; opts.device = -1; opts.flags = flags;
  uint8_t* out = nullptr; size_t out_n = 0;
  static const uint8_t dummy = 0;
  if (!p) p = &dummy;
  int rc;
  if (KIND == 0) rc = api.bzip2_compress(p, n, arg, &out, &out_n, nullptr);
  else if (KIND == 1) rc = api.bzip2_decompress(p, n, arg, &out, &out_n, nullptr);
  else if (KIND == 2) rc = api.bwtc_compress(p, n, arg, &out, &out_n, flags ? &opts : nullptr);
  else rc = api.bwtc_decompress(p, n, &out, &out_n, nullptr);
  if (rc != 0) return throw_code(env, rc);
  return wrap_result(env, out, out_n);
}

// bzip2Table(input, multistream) -> Float64Array [pos0, size0, pos1, size1, ...]   (Bzip2.table)
napi_value bzip2_table(napi_env env, napi_callback_info info) {
  if (!load_api()) { napi_throw_error(env, nullptr, api.error.c_str()); return nullptr; }
  size_t argc = 2; napi_value argv[2];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  const uint8_t* p = nullptr; size_t n = 0;
  if (argc < 1 || !get_bytes(env, argv[0], &p, &n)) { napi_throw_type_error(env, nullptr, "expected a Uint8Array or Buffer"); return nullptr; }
  int32_t multi = 0;
  if (argc >= 2) napi_get_value_int32(env, argv[1], &multi);
  long cap = (long)(n / 32 + 64);
  uint64_t* pos = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)cap);
  uint32_t* size = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)cap);
  static const uint8_t dummy = 0;
  long nb = api.bzip2_table(p ? p : &dummy, n, multi, pos, size, cap, nullptr);
  if (nb < 0) { free(pos); free(size); return throw_code(env, (int)nb); }
  if (nb > cap) nb = cap;
  napi_value ab, ta; void* dst;
  napi_create_arraybuffer(env, sizeof(double) * 2 * (size_t)nb, &dst, &ab);
  for (long i = 0; i < nb; i++) { ((double*)dst)[2 * i] = (double)pos[i]; ((double*)dst)[2 * i + 1] = (double)size[i]; }
  free(pos); free(size);
  napi_create_typedarray(env, napi_float64_array, 2 * (size_t)nb, ab, 0, &ta);
  return ta;
}
// bzip2DecompressBlock(input, bitpos) -> Uint8Array   (Bzip2.decompressBlock)
napi_value bzip2_block(napi_env env, napi_callback_info info) {
  if (!load_api()) { napi_throw_error(env, nullptr, api.error.c_str()); return nullptr; }
  size_t argc = 2; napi_value argv[2];
  napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
  const uint8_t* p = nullptr; size_t n = 0;
  if (argc < 2 || !get_bytes(env, argv[0], &p, &n)) { napi_throw_type_error(env, nullptr, "expected (Uint8Array, bit position)"); return nullptr; }
  double bit = 0; napi_get_value_double(env, argv[1], &bit);
  uint8_t* out = nullptr; size_t out_n = 0;
  static const uint8_t dummy = 0;
  const int rc = api.bzip2_decompress_block(p ? p : &dummy, n, (uint64_t)bit, &out, &out_n, nullptr);
  if (rc != 0) return throw_code(env, rc);
  return wrap_result(env, out, out_n);
}

napi_value device_count(napi_env env, napi_callback_info) {
  if (!load_api()) { napi_throw_error(env, nullptr, api.error.c_str()); return nullptr; }
  napi_value v; napi_create_int32(env, api.device_count(), &v); return v;
}
napi_value version(napi_env env, napi_callback_info) {
  if (!load_api()) { napi_throw_error(env, nullptr, api.error.c_str()); return nullptr; }
  napi_value v; napi_create_string_utf8(env, api.version(), NAPI_AUTO_LENGTH, &v); return v;
}

napi_value trim(napi_env env, napi_callback_info) {          // releases the workspace the library keeps between compress calls
  if (!load_api()) { napi_throw_error(env, nullptr, api.error.c_str()); return nullptr; }
  api.trim();
  napi_value v; napi_get_undefined(env, &v); return v;
}

napi_value init(napi_env env, napi_value exports) {
  napi_property_descriptor props[] = {
    {"bzip2Compress", nullptr, call_stream<0>, nullptr, nullptr, nullptr, napi_default, nullptr},
    {"bzip2Decompress", nullptr, call_stream<1>, nullptr, nullptr, nullptr, napi_default, nullptr},
    {"bwtcCompress", nullptr, call_stream<2>, nullptr, nullptr, nullptr, napi_default, nullptr},
    {"bwtcDecompress", nullptr, call_stream<3>, nullptr, nullptr, nullptr, napi_default, nullptr},
    {"bzip2Table", nullptr, bzip2_table, nullptr, nullptr, nullptr, napi_default, nullptr},
    {"bzip2DecompressBlock", nullptr, bzip2_block, nullptr, nullptr, nullptr, napi_default, nullptr},
    {"deviceCount", nullptr, device_count, nullptr, nullptr, nullptr, napi_default, nullptr},
    {"version", nullptr, version, nullptr, nullptr, nullptr, napi_default, nullptr},
    {"trim", nullptr, trim, nullptr, nullptr, nullptr, napi_default, nullptr},
  };
  napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props);
  return exports;
}

}  // namespace

NAPI_MODULE(cjs_napi, init)
