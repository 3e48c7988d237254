// Packed-weights file reader (host code only; no kernel in this file).
//
// Every entry point of include/cube_hip.h reads weights in layouts that the load-time packer produces from a reference
// state_dict (cube_amd/networks/packing.py: BatchNorm folded into the gated convolutions, gate-column order, chunked K,
// int8x2 slices and scales, bias tables, LSTM gate-row orders, MFMA-fragment order of W_hh ...).  That packer is Python.
// A host in another language gets the SAME bytes from a file written once, offline, by `packing.save_packed`
// (`Parser.save_packed(path)`): this reader maps the file and hands out, by name, each tensor's host pointer, type and
// shape, and the integer scalars that go with it (Kc, N, half, the column offsets of the concat buffers).  The host
// copies the tensors to device memory it owns (hipMemcpy) and fills cube_gemm_args etc. from the scalars.
//
// File layout (little endian):
//   char magic[8] = "CUBEPACK"; u32 version = 1; u32 n_entries; u64 table_bytes; u64 data_offset (multiple of 256)
//   n_entries x { u16 name_len; char name[name_len]; u8 dtype; u8 ndim; i64 dims[ndim]; u64 offset; u64 nbytes }
//   data: each tensor at data_offset + offset, 256-byte aligned; integer scalars are dtype CUBE_PACK_I64, ndim 0.
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <vector>

#include "common.h"
#include "../../include/cube_hip.h"

struct cube_packfile {
  void* map = nullptr;
  size_t size = 0;
  std::vector<std::string> names;
  std::vector<cube_pack_entry> entries;
};

namespace {

const int ELEM_BYTES[] = {4, 2, 1, 4, 8, 1};  // f32, bf16, i8, i32, i64, u8

struct Cursor {
  const unsigned char* p;
  const unsigned char* end;
  bool ok = true;
  template <class T> T get() {
    T v{};
    if (!ok || (size_t)(end - p) < sizeof(T)) { ok = false; return v; }
    memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
};

}  // namespace

extern "C" int cube_packfile_open(const char* path, cube_packfile** out) {
  if (!path || !out) CUBE_FAIL(CUBE_ERR_ARG, "cube_packfile_open: null argument");
  *out = nullptr;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) CUBE_FAIL(CUBE_ERR_ARG, "cube_packfile_open: cannot open %s", path);
  struct stat sb;
  if (fstat(fd, &sb) != 0 || sb.st_size < 32) {
    close(fd);
    CUBE_FAIL(CUBE_ERR_ARG, "cube_packfile_open: %s is not a packed-weights file (too short)", path);
  }
  void* map = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) CUBE_FAIL(CUBE_ERR_ARG, "cube_packfile_open: mmap of %s failed", path);
  cube_packfile* f = new cube_packfile;
  f->map = map;
  f->size = (size_t)sb.st_size;
  const unsigned char* base = (const unsigned char*)map;
  auto fail = [&](const char* why) {
    cube_set_error("cube_packfile_open: %s: %s", path, why);
    munmap(f->map, f->size);
    delete f;
    return CUBE_ERR_ARG;
  };
  if (memcmp(base, "CUBEPACK", 8) != 0) return fail("bad magic");
  Cursor c{base + 8, base + f->size};
  const uint32_t version = c.get<uint32_t>(), n = c.get<uint32_t>();
  const uint64_t table_bytes = c.get<uint64_t>(), data_off = c.get<uint64_t>();
  if (version != 1) return fail("unknown version");
  if (data_off % 256 || data_off > f->size || table_bytes > f->size || 32 + table_bytes > data_off) return fail("bad header");
  c.end = base + 32 + table_bytes;
  f->names.reserve(n);
  f->entries.reserve(n);
  for (uint32_t i = 0; i < n; ++i) {
    const uint16_t nl = c.get<uint16_t>();
    if (!c.ok || (size_t)(c.end - c.p) < nl) return fail("truncated entry table");
    f->names.emplace_back((const char*)c.p, nl);
    c.p += nl;
    cube_pack_entry e;
    memset(&e, 0, sizeof(e));
    e.dtype = c.get<uint8_t>();
    e.ndim = c.get<uint8_t>();
    if (!c.ok || e.dtype > CUBE_PACK_U8 || e.ndim > CUBE_PACK_MAX_DIMS) return fail("bad entry type / rank");
    uint64_t count = 1;
    for (int d = 0; d < e.ndim; ++d) {
      e.dims[d] = c.get<int64_t>();
      if (e.dims[d] < 0) return fail("negative dimension");
      count *= (uint64_t)e.dims[d];
    }
    const uint64_t off = c.get<uint64_t>(), nbytes = c.get<uint64_t>();
    if (!c.ok) return fail("truncated entry table");
    if (nbytes != count * ELEM_BYTES[e.dtype] || off % 256 || data_off + off > f->size || nbytes > f->size - data_off - off)
      return fail("entry outside the file");
    e.nbytes = nbytes;
    e.data = base + data_off + off;
    f->entries.push_back(e);
  }
  for (uint32_t i = 0; i < n; ++i) f->entries[i].name = f->names[i].c_str();
  *out = f;
  return CUBE_OK;
}

extern "C" int cube_packfile_count(const cube_packfile* f) { return f ? (int)f->entries.size() : 0; }

extern "C" int cube_packfile_entry(const cube_packfile* f, int i, cube_pack_entry* out) {
  if (!f || !out || i < 0 || i >= (int)f->entries.size()) CUBE_FAIL(CUBE_ERR_ARG, "cube_packfile_entry: bad index %d", i);
  *out = f->entries[i];
  return CUBE_OK;
}

extern "C" int cube_packfile_find(const cube_packfile* f, const char* name, cube_pack_entry* out) {
  if (!f || !name || !out) CUBE_FAIL(CUBE_ERR_ARG, "cube_packfile_find: null argument");
  for (size_t i = 0; i < f->entries.size(); ++i)
    if (f->names[i] == name) {
      *out = f->entries[i];
      return CUBE_OK;
    }
  CUBE_FAIL(CUBE_ERR_ARG, "cube_packfile_find: no entry named %s", name);
}

extern "C" int cube_packfile_int(const cube_packfile* f, const char* name, long long* value) {
  cube_pack_entry e;
  if (!value) CUBE_FAIL(CUBE_ERR_ARG, "cube_packfile_int: null argument");
  const int rc = cube_packfile_find(f, name, &e);
  if (rc != CUBE_OK) return rc;
  if (e.dtype != CUBE_PACK_I64 || e.ndim != 0) CUBE_FAIL(CUBE_ERR_ARG, "cube_packfile_int: %s is not an integer scalar", name);
  int64_t v;
  memcpy(&v, e.data, 8);
  *value = v;
  return CUBE_OK;
}

extern "C" void cube_packfile_close(cube_packfile* f) {
  if (!f) return;
  if (f->map) munmap(f->map, f->size);
  delete f;
}
