// image.hip -- tape images: a model forward compiled ahead of time into ONE relocatable file, so that a host without the
// Python graph compiler (audioeditingcode_amd/unet.py, codec.py, tape.py) can run it through the C ABI alone.
//
// The Python side (audioeditingcode_amd/image.py) lays every buffer an op tape references -- weights, tables, activations,
// inputs, outputs -- into one arena, rewrites the ops' device pointers as arena offsets and writes
//     header | programs (name, first op, op count) | named buffers (name, offset, bytes) | ops | arena snapshot.
// aed_image_load() allocates the arena on the device, uploads the snapshot and relocates the pointers;
// aed_image_run(image, "forward", stream) is then exactly aed_tape_run() on the relocated ops.  This is the model-level
// boundary SURVEY 8(b) sketched (aed_create / aed_unet_forward / ...), realised as "load a compiled model, fill its named
// inputs, run a named program".
#include <stdio.h>
#include <exception>
#include <string.h>
#include <string>
#include <vector>
#include "aed_common.h"

namespace {
constexpr uint64_t kNull = ~0ull;
struct Header {
    char magic[8];          // "AEDIMG1\0"
    uint32_t version, n_ops, n_programs, n_names;
    uint64_t arena_bytes, snapshot_bytes;       // snapshot covers arena[0, snapshot_bytes)
};
struct Program { char name[48]; uint32_t first, count; };
struct Named { char name[48]; uint64_t offset, nbytes; };
struct Image {
    char* arena = nullptr;
    bool host = false;
    uint64_t arena_bytes = 0;
    std::vector<aed_op> ops;
    std::vector<Program> programs;
    std::vector<Named> names;
};
bool read_all(FILE* f, void* dst, size_t n) { return n == 0 || fread(dst, 1, n, f) == n; }
}  // namespace

extern "C" {

int aed_image_load(const char* path, int flags, void** image_out) {
    AED_REQUIRE(path && image_out, "aed_image_load: null argument");
    FILE* f = fopen(path, "rb");
    AED_REQUIRE(f != nullptr, "aed_image_load: cannot open %s", path);
    Header h;
    Image* im = new Image();
    int rc = 1;
    do {
        if (!read_all(f, &h, sizeof(h)) || memcmp(h.magic, "AEDIMG1", 8) != 0 || h.version != 1) {
            aed_set_error("aed_image_load: %s is not a version-1 tape image", path);
            break;
        }
        if (h.snapshot_bytes > h.arena_bytes) { aed_set_error("aed_image_load: corrupt header"); break; }
        // the three tables and the snapshot must fit the file: a corrupt count must not turn into a huge resize()
        long fsize = -1;
        if (fseek(f, 0, SEEK_END) == 0) fsize = ftell(f);
        if (fsize < 0 || fseek(f, (long)sizeof(h), SEEK_SET) != 0) { aed_set_error("aed_image_load: cannot size %s", path); break; }
        const uint64_t room = (uint64_t)fsize - sizeof(h);
        if (h.n_programs > room / sizeof(Program) || h.n_names > room / sizeof(Named) || h.n_ops > room / sizeof(aed_op) ||
            sizeof(Program) * (uint64_t)h.n_programs + sizeof(Named) * (uint64_t)h.n_names + sizeof(aed_op) * (uint64_t)h.n_ops +
                    h.snapshot_bytes > room) {
            aed_set_error("aed_image_load: %s: header counts exceed the file size", path);
            break;
        }
        try {
            im->programs.resize(h.n_programs);
            im->names.resize(h.n_names);
            im->ops.resize(h.n_ops);
        } catch (const std::exception&) {       // never let an allocation failure cross the C boundary
            aed_set_error("aed_image_load: out of host memory (tables)");
            break;
        }
        if (!read_all(f, im->programs.data(), sizeof(Program) * h.n_programs) ||
            !read_all(f, im->names.data(), sizeof(Named) * h.n_names) ||
            !read_all(f, im->ops.data(), sizeof(aed_op) * h.n_ops)) {
            aed_set_error("aed_image_load: %s is truncated (tables)", path);
            break;
        }
        im->arena_bytes = h.arena_bytes;
        im->host = (flags & 1) != 0;            // inspection / tests: keep the arena in host memory (not runnable)
        std::vector<char> chunk;
        try {
            chunk.resize(std::min<uint64_t>(h.snapshot_bytes ? h.snapshot_bytes : 1, 64ull << 20));
        } catch (const std::exception&) { aed_set_error("aed_image_load: out of host memory (staging)"); break; }
        if (im->host) {
            im->arena = (char*)calloc(h.arena_bytes ? h.arena_bytes : 1, 1);
            if (!im->arena) { aed_set_error("aed_image_load: out of host memory"); break; }
        } else {
            if (hipMalloc((void**)&im->arena, h.arena_bytes ? h.arena_bytes : 1) != hipSuccess) {
                im->arena = nullptr;
                aed_set_error("aed_image_load: hipMalloc of %llu bytes failed", (unsigned long long)h.arena_bytes);
                break;
            }
            if (hipMemset(im->arena, 0, h.arena_bytes) != hipSuccess) { aed_set_error("aed_image_load: hipMemset failed"); break; }
        }
        bool ok = true;
        for (uint64_t off = 0; off < h.snapshot_bytes && ok; off += chunk.size()) {
            const size_t n = (size_t)std::min<uint64_t>(chunk.size(), h.snapshot_bytes - off);
            ok = read_all(f, chunk.data(), n);
            if (!ok) break;
            if (im->host) memcpy(im->arena + off, chunk.data(), n);
            else ok = hipMemcpy(im->arena + off, chunk.data(), n, hipMemcpyHostToDevice) == hipSuccess;
        }
        if (!ok) { aed_set_error("aed_image_load: %s is truncated (arena snapshot) or the upload failed", path); break; }
        // relocate
        bool bad = false;
        for (auto& op : im->ops)
            for (int k = 0; k < 10; ++k) {
                const uint64_t off = (uint64_t)(uintptr_t)op.p[k];
                if (off == kNull) op.p[k] = nullptr;
                else if (off >= h.arena_bytes) bad = true;
                else op.p[k] = im->arena + off;
            }
        for (auto& p : im->programs) bad |= (uint64_t)p.first + p.count > h.n_ops;
        for (auto& n : im->names) bad |= n.nbytes > h.arena_bytes || n.offset > h.arena_bytes - n.nbytes;   // no u64 wrap
        if (bad) { aed_set_error("aed_image_load: %s has out-of-range offsets", path); break; }
        rc = 0;
    } while (false);
    fclose(f);
    if (rc) {
        if (im->arena) { if (im->host) free(im->arena); else (void)hipFree(im->arena); }
        delete im;
        return rc;
    }
    *image_out = im;
    return 0;
}

int aed_image_free(void* image) {
    Image* im = (Image*)image;
    if (!im) return 0;
    if (im->arena) { if (im->host) free(im->arena); else AED_CHECK_HIP(hipFree(im->arena)); }
    delete im;
    return 0;
}

static const Program* find_program(const Image* im, const char* name) {
    for (auto& p : im->programs)
        if (strncmp(p.name, name, sizeof(p.name)) == 0) return &p;
    return nullptr;
}
static const Named* find_named(const Image* im, const char* name) {
    for (auto& n : im->names)
        if (strncmp(n.name, name, sizeof(n.name)) == 0) return &n;
    return nullptr;
}

int aed_image_run(void* image, const char* program, void* stream) {
    Image* im = (Image*)image;
    AED_REQUIRE(im && program, "aed_image_run: null argument");
    AED_REQUIRE(!im->host, "aed_image_run: this image was loaded into host memory (flags & 1): inspection only");
    const Program* p = find_program(im, program);
    AED_REQUIRE(p != nullptr, "aed_image_run: no program '%s' in this image", program);
    return aed_tape_run(im->ops.data() + p->first, (int)p->count, stream);
}

int aed_image_program(void* image, const char* program, const aed_op** ops_out, int* n_out) {
    Image* im = (Image*)image;
    AED_REQUIRE(im && program && ops_out && n_out, "aed_image_program: null argument");
    const Program* p = find_program(im, program);
    AED_REQUIRE(p != nullptr, "aed_image_program: no program '%s' in this image", program);
    *ops_out = im->ops.data() + p->first;
    *n_out = (int)p->count;
    return 0;
}

int aed_image_buffer(void* image, const char* name, void** ptr_out, uint64_t* nbytes_out) {
    Image* im = (Image*)image;
    AED_REQUIRE(im && name && ptr_out, "aed_image_buffer: null argument");
    const Named* n = find_named(im, name);
    AED_REQUIRE(n != nullptr, "aed_image_buffer: no buffer '%s' in this image", name);
    *ptr_out = im->arena + n->offset;
    if (nbytes_out) *nbytes_out = n->nbytes;
    return 0;
}

static int image_copy(void* image, const char* name, void* host, uint64_t nbytes, void* stream, bool in) {
    Image* im = (Image*)image;
    AED_REQUIRE(im && name && host, "aed_image_copy: null argument");
    const Named* n = find_named(im, name);
    AED_REQUIRE(n != nullptr, "aed_image_copy: no buffer '%s' in this image", name);
    AED_REQUIRE(nbytes <= n->nbytes, "aed_image_copy: %llu bytes do not fit buffer '%s' (%llu bytes)",
                (unsigned long long)nbytes, name, (unsigned long long)n->nbytes);
    char* dev = im->arena + n->offset;
    if (im->host) {
        if (in) memcpy(dev, host, nbytes); else memcpy(host, dev, nbytes);
        return 0;
    }
    AED_CHECK_HIP(hipMemcpyAsync(in ? (void*)dev : host, in ? host : (void*)dev, nbytes,
                                 in ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, (hipStream_t)stream));
    AED_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int aed_image_copy_in(void* image, const char* name, const void* host, uint64_t nbytes, void* stream) {
    return image_copy(image, name, (void*)host, nbytes, stream, true);
}
int aed_image_copy_out(void* image, const char* name, void* host, uint64_t nbytes, void* stream) {
    return image_copy(image, name, host, nbytes, stream, false);
}

}  // extern "C"
