// Host-side asset I/O above the tracing boundary (include/rrt_host.h):
// OBJ reader with obj-rs 0.6.2 semantics, Model::new's triangle construction,
// minimal PNG decode/encode on zlib.  No GPU code here.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rrt_host.h"
#include "host_io.hpp"

namespace rrt {

thread_local std::string g_host_err;

static int hfail(int code, const std::string& m);
// no C++ exception crosses the C ABI (see rrt_api.cpp)
static int host_on_exception() {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        return hfail(RRT_E_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return hfail(RRT_E_STATE, std::string("internal error: ") + e.what());
    } catch (...) {
        return hfail(RRT_E_STATE, "internal error");
    }
}
static int hfail(int code, const std::string& m) {
    g_host_err = m;
    set_last_error(m);
    return code;
}

// ---------------------------------------------------------------------------
// OBJ
// ---------------------------------------------------------------------------
struct ObjData {
    std::vector<float> pos, nrm;          // 3 floats each
    std::vector<int32_t> face_p, face_n;  // 3 indices per face
};

static bool parse_index_triplet(const char* tok, long* p, long* t, long* n, bool* has_n) {
    // forms: p | p/t | p//n | p/t/n
    char* end = nullptr;
    *p = std::strtol(tok, &end, 10);
    if (end == tok) return false;
    *t = 0;
    *n = 0;
    *has_n = false;
    if (*end != '/') return *end == '\0';
    ++end;
    if (*end != '/') {
        char* e2 = nullptr;
        *t = std::strtol(end, &e2, 10);
        if (e2 == end) return false;
        end = e2;
        if (*end == '\0') return true;
        if (*end != '/') return false;
    }
    ++end;
    char* e3 = nullptr;
    *n = std::strtol(end, &e3, 10);
    if (e3 == end) return false;
    *has_n = true;
    return *e3 == '\0';
}

static int read_obj(const char* path, ObjData* out, bool triangulate) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return hfail(RRT_E_IO, std::string("cannot open ") + path);
    std::string line;
    char buf[4096];
    int lineno = 0;
    int rc = RRT_OK;
    while (rc == RRT_OK && std::fgets(buf, sizeof buf, f)) {
        ++lineno;
        line.assign(buf);
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.resize(hash);
        std::vector<std::string> tok;
        size_t i = 0;
        while (i < line.size()) {
            while (i < line.size() && (line[i] == ' ' || line[i] == '\t')) ++i;
            size_t j = i;
            while (j < line.size() && line[j] != ' ' && line[j] != '\t') ++j;
            if (j > i) tok.push_back(line.substr(i, j - i));
            i = j;
        }
        if (tok.empty()) continue;
        if (tok[0] == "v" || tok[0] == "vn") {
            if (tok.size() < 4) {
                rc = hfail(RRT_E_FORMAT, std::string(path) + ":" + std::to_string(lineno) + ": short vertex line");
                break;
            }
            std::vector<float>& dst = tok[0] == "v" ? out->pos : out->nrm;
            for (int k = 1; k <= 3; ++k) dst.push_back(std::strtof(tok[k].c_str(), nullptr));  // f32, as obj-rs
        } else if (tok[0] == "f") {
            if (tok.size() < 4 || (tok.size() != 4 && !triangulate)) {
                rc = hfail(RRT_E_FORMAT, std::string(path) + ":" + std::to_string(lineno) +
                                             ": face is not a triangle (obj-rs load_obj refuses untriangulated models; "
                                             "RRT_OBJ_TRIANGULATE fans them)");
                break;
            }
            std::vector<int32_t> cp, cn;      // the face's corners
            for (size_t k = 1; k < tok.size(); ++k) {
                long p, t, n;
                bool has_n;
                if (!parse_index_triplet(tok[k].c_str(), &p, &t, &n, &has_n) || !has_n) {
                    rc = hfail(RRT_E_FORMAT, std::string(path) + ":" + std::to_string(lineno) +
                                                 ": face corner needs position and normal indices");
                    break;
                }
                const long np = (long)out->pos.size() / 3, nn = (long)out->nrm.size() / 3;
                p = p > 0 ? p - 1 : np + p;
                n = n > 0 ? n - 1 : nn + n;
                if (p < 0 || p >= np || n < 0 || n >= nn) {
                    rc = hfail(RRT_E_FORMAT, std::string(path) + ":" + std::to_string(lineno) + ": index out of range");
                    break;
                }
                cp.push_back((int32_t)p);
                cn.push_back((int32_t)n);
            }
            if (rc != RRT_OK) break;
            // a triangle as it stands; a polygon (EXTENSION, RRT_OBJ_TRIANGULATE) as the fan (c0, ci, ci+1)
            for (size_t i = 1; i + 1 < cp.size(); ++i) {
                const size_t tri[3] = {0, i, i + 1};
                for (size_t c : tri) {
                    out->face_p.push_back(cp[c]);
                    out->face_n.push_back(cn[c]);
                }
            }
        }
        // o, g, s, vt, usemtl, mtllib: irrelevant to Vertex{position, normal}
    }
    std::fclose(f);
    return rc;
}

static void norm3(const double in[3], double out[3]) {   // cgmath normalize
    const double m2 = in[0] * in[0] + in[1] * in[1] + in[2] * in[2];
    const double inv = 1.0 / std::sqrt(m2);
    for (int a = 0; a < 3; ++a) out[a] = in[a] * inv;
}

int load_model(const char* path, const double position[3], const double scale[3], std::vector<double>* verts,
               std::vector<double>* normals, uint32_t flags) {
    ObjData o;
    int rc = read_obj(path, &o, (flags & RRT_OBJ_TRIANGULATE) != 0);
    if (rc) return rc;
    const long n_idx = (long)o.face_p.size();
    if (n_idx < 5) return hfail(RRT_E_FORMAT, std::string(path) + ": fewer than two faces (model.rs:31 would underflow)");
    verts->clear();
    normals->clear();
    // for i in (0..indices.len()-4).step_by(3)   (model.rs:31)
    for (long i = 0; i < n_idx - 4; i += 3) {
        double nsum[3] = {0, 0, 0};
        for (int c = 0; c < 3; ++c) {
            const float* p = &o.pos[3 * (size_t)o.face_p[(size_t)(i + c)]];
            for (int a = 0; a < 3; ++a) {
                // Matrix4 (T*S) * point: fl(fl(s*x) + t)   (model.rs:29-41)
                const double scaled = scale[a] * (double)p[a];
                verts->push_back(scaled + position[a]);
            }
            const float* nf = &o.nrm[3 * (size_t)o.face_n[(size_t)(i + c)]];
            const double nd[3] = {(double)nf[0], (double)nf[1], (double)nf[2]};
            double nn[3];
            norm3(nd, nn);   // vertex2normal (model.rs:97-103)
            if (c == 0) for (int a = 0; a < 3; ++a) nsum[a] = nn[a];
            else for (int a = 0; a < 3; ++a) nsum[a] = nsum[a] + nn[a];
        }
        double nrm[3];
        norm3(nsum, nrm);    // (n0 + n1 + n2).normalize()  (model.rs:42)
        for (int a = 0; a < 3; ++a) normals->push_back(nrm[a]);
    }
    return RRT_OK;
}

// ---------------------------------------------------------------------------
// PNG
// ---------------------------------------------------------------------------
static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

int load_png_rgb8(const char* path, std::vector<uint8_t>* rgb, uint32_t* w_out, uint32_t* h_out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return hfail(RRT_E_IO, std::string("cannot open ") + path);
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    std::fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return hfail(RRT_E_FORMAT, std::string(path) + ": not a PNG");
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    size_t at = 8;
    bool end = false;
    while (!end && at + 12 <= file.size()) {
        const uint32_t len = be32(&file[at]);
        const char* type = (const char*)&file[at + 4];
        if (at + 12 + (size_t)len > file.size()) return hfail(RRT_E_FORMAT, std::string(path) + ": truncated chunk");
        const uint8_t* data = &file[at + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) return hfail(RRT_E_FORMAT, std::string(path) + ": bad IHDR");
            w = be32(data);
            h = be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            end = true;
        }
        at += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || ctype < 0) return hfail(RRT_E_FORMAT, std::string(path) + ": missing IHDR");
    // the PNG limit (2^31 - 1 per side) and the renderer's own (plan_frame: 2^31 - 1 pixels): beyond it the scanline
    // and image sizes below (up to 8 bytes a pixel) could wrap size_t, and no frame or cubemap face is that large
    if (w > 0x7fffffffu || h > 0x7fffffffu || (uint64_t)w * h > 0x7fffffffull)
        return hfail(RRT_E_FORMAT, std::string(path) + ": image of " + std::to_string(w) + " x " + std::to_string(h) + " pixels is too large");
    if (interlace > 1) return hfail(RRT_E_FORMAT, std::string(path) + ": unknown interlace method");
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: return hfail(RRT_E_FORMAT, std::string(path) + ": bad colour type");
    }
    const bool low = depth == 1 || depth == 2 || depth == 4;
    if (!(depth == 8 || depth == 16 || (low && (ctype == 0 || ctype == 3))) || (ctype == 3 && depth == 16))
        return hfail(RRT_E_FORMAT, std::string(path) + ": bit depth " + std::to_string(depth) + " is not valid for colour type " + std::to_string(ctype));
    const size_t bits_pp = (size_t)channels * depth;
    const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;      // filter distance in bytes
    // the image, or its seven Adam7 sub-images (x0, y0, dx, dy), each a run of filtered scanlines of its own
    static const uint32_t kPass[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    struct Pass { uint32_t x0, y0, dx, dy, pw, ph; size_t stride; };
    std::vector<Pass> passes;
    size_t raw_size = 0;
    for (int k = 0; k < (interlace ? 7 : 1); ++k) {
        Pass ps;
        if (interlace) { ps.x0 = kPass[k][0]; ps.y0 = kPass[k][1]; ps.dx = kPass[k][2]; ps.dy = kPass[k][3]; }
        else { ps.x0 = ps.y0 = 0; ps.dx = ps.dy = 1; }
        ps.pw = w > ps.x0 ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0;
        ps.ph = h > ps.y0 ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
        if (ps.pw == 0 || ps.ph == 0) continue;            // an empty pass has no scanlines at all
        ps.stride = ((size_t)ps.pw * bits_pp + 7) / 8;
        raw_size += (ps.stride + 1) * (size_t)ps.ph;
        passes.push_back(ps);
    }
    std::vector<uint8_t> raw(raw_size);
    uLongf raw_len = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || raw_len != raw.size()) return hfail(RRT_E_FORMAT, std::string(path) + ": zlib inflate failed");
    rgb->assign((size_t)w * h * 3, 0);
    // image 0.23.14 `to_rgb8`: alpha dropped, palettes expanded, grey replicated, packed grey scaled to 0..255, and a
    // 16-bit sample reduced to its HIGH byte (color.rs downcast_channel: c16 >> 8) -- no 16-bit asset exists in the
    // reference, so that last rule is restated from the crate's source as remembered: parity unpinned.
    const auto sample8 = [&](const uint8_t* row, size_t idx) -> uint32_t {      // idx-th sample of a scanline
        if (depth == 8) return row[idx];
        if (depth == 16) return row[2 * idx];
        const uint32_t per = 8u / (uint32_t)depth, sh = (per - 1 - (uint32_t)(idx % per)) * (uint32_t)depth;
        return (row[idx / per] >> sh) & ((1u << depth) - 1u);
    };
    size_t at_raw = 0;
    std::vector<uint8_t> prev, cur;
    for (const Pass& ps : passes) {
        prev.assign(ps.stride, 0);
        cur.assign(ps.stride, 0);
        for (uint32_t py = 0; py < ps.ph; ++py) {
            const uint8_t ft = raw[at_raw];
            const uint8_t* in = &raw[at_raw + 1];
            at_raw += ps.stride + 1;
            for (size_t i = 0; i < ps.stride; ++i) {
                const int a = i >= bpp ? cur[i - bpp] : 0;
                const int b = prev[i];
                const int c = i >= bpp ? prev[i - bpp] : 0;
                int v = in[i];
                switch (ft) {
                    case 0: break;
                    case 1: v += a; break;
                    case 2: v += b; break;
                    case 3: v += (a + b) / 2; break;
                    case 4: v += paeth(a, b, c); break;
                    default: return hfail(RRT_E_FORMAT, std::string(path) + ": bad filter type");
                }
                cur[i] = (uint8_t)v;
            }
            const uint32_t y = ps.y0 + py * ps.dy;
            for (uint32_t px = 0; px < ps.pw; ++px) {
                const uint32_t x = ps.x0 + px * ps.dx;
                uint32_t r, g, b;
                if (ctype == 2 || ctype == 6) {
                    r = sample8(cur.data(), (size_t)channels * px); g = sample8(cur.data(), (size_t)channels * px + 1); b = sample8(cur.data(), (size_t)channels * px + 2);
                } else if (ctype == 0 || ctype == 4) {
                    r = sample8(cur.data(), (size_t)channels * px);
                    if (low) r = r * 255u / ((1u << depth) - 1u);
                    g = b = r;
                } else {
                    const uint32_t idx = sample8(cur.data(), px);
                    if (3 * (size_t)idx + 2 >= plte.size()) return hfail(RRT_E_FORMAT, std::string(path) + ": palette index out of range");
                    r = plte[3 * idx]; g = plte[3 * idx + 1]; b = plte[3 * idx + 2];
                }
                uint8_t* dst = &(*rgb)[((size_t)y * w + x) * 3];
                dst[0] = (uint8_t)r; dst[1] = (uint8_t)g; dst[2] = (uint8_t)b;
            }
            prev.swap(cur);
        }
    }
    *w_out = w;
    *h_out = h;
    return RRT_OK;
}

static void put_chunk(std::vector<uint8_t>* out, const char* type, const uint8_t* data, size_t len) {
    const uint8_t l[4] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len};
    out->insert(out->end(), l, l + 4);
    const size_t start = out->size();
    out->insert(out->end(), type, type + 4);
    if (len) out->insert(out->end(), data, data + len);
    const uint32_t crc = (uint32_t)crc32(0L, out->data() + start, (uInt)(len + 4));
    const uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    out->insert(out->end(), c, c + 4);
}

int save_png_rgb8(const char* path, const uint8_t* rgb, uint32_t w, uint32_t h) {
    if (!rgb || w == 0 || h == 0) return hfail(RRT_E_INVALID, "bad image");
    std::vector<uint8_t> raw(((size_t)w * 3 + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw[((size_t)w * 3 + 1) * y] = 0;
        std::memcpy(&raw[((size_t)w * 3 + 1) * y + 1], rgb + (size_t)y * w * 3, (size_t)w * 3);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return hfail(RRT_E_IO, "zlib deflate failed");
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                        (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 2, 0, 0, 0};
    put_chunk(&out, "IHDR", ihdr, 13);
    put_chunk(&out, "IDAT", z.data(), zlen);
    put_chunk(&out, "IEND", nullptr, 0);
    FILE* f = std::fopen(path, "wb");
    if (!f) return hfail(RRT_E_IO, std::string("cannot write ") + path);
    const size_t wr = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    if (wr != out.size()) return hfail(RRT_E_IO, std::string("short write to ") + path);
    return RRT_OK;
}

}  // namespace rrt

extern "C" {

int rrt_host_load_model(const char* obj_path, const double position[3], const double scale[3], double** verts,
                        double** normals, int64_t* n_triangles) {
    return rrt_host_load_model_ex(obj_path, position, scale, 0u, verts, normals, n_triangles);
}

int rrt_host_load_model_ex(const char* obj_path, const double position[3], const double scale[3], uint32_t flags,
                           double** verts, double** normals, int64_t* n_triangles) try {
    if (!obj_path || !position || !scale || !verts || !normals || !n_triangles) return rrt::hfail(RRT_E_INVALID, "null argument");
    std::vector<double> v, n;
    const int rc = rrt::load_model(obj_path, position, scale, &v, &n, flags);
    if (rc) return rc;
    *n_triangles = (int64_t)(n.size() / 3);
    *verts = (double*)std::malloc(v.size() * sizeof(double) + 8);
    *normals = (double*)std::malloc(n.size() * sizeof(double) + 8);
    if (!*verts || !*normals) return rrt::hfail(RRT_E_NOMEM, "out of host memory");
    std::memcpy(*verts, v.data(), v.size() * sizeof(double));
    std::memcpy(*normals, n.data(), n.size() * sizeof(double));
    return RRT_OK;
} catch (...) {
    return rrt::host_on_exception();
}

int rrt_host_add_model(rrt_scene* s, const char* obj_path, int material, const double position[3], const double scale[3]) {
    return rrt_host_add_model_ex(s, obj_path, material, position, scale, 0u);
}

int rrt_host_add_model_ex(rrt_scene* s, const char* obj_path, int material, const double position[3], const double scale[3],
                          uint32_t flags) try {
    if (!s || !obj_path || !position || !scale) return rrt::hfail(RRT_E_INVALID, "null argument");
    std::vector<double> v, n;
    const int rc = rrt::load_model(obj_path, position, scale, &v, &n, flags);
    if (rc) return rc;
    return rrt_add_mesh(s, v.data(), n.data(), (int64_t)(n.size() / 3), material);
} catch (...) {
    return rrt::host_on_exception();
}

int rrt_host_load_png_rgb8(const char* path, uint8_t** rgb8, uint32_t* w, uint32_t* h) try {
    if (!path || !rgb8 || !w || !h) return rrt::hfail(RRT_E_INVALID, "null argument");
    std::vector<uint8_t> px;
    const int rc = rrt::load_png_rgb8(path, &px, w, h);
    if (rc) return rc;
    *rgb8 = (uint8_t*)std::malloc(px.size() + 8);
    if (!*rgb8) return rrt::hfail(RRT_E_NOMEM, "out of host memory");
    std::memcpy(*rgb8, px.data(), px.size());
    return RRT_OK;
} catch (...) {
    return rrt::host_on_exception();
}

int rrt_host_save_png_rgb8(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h) try {
    if (!path) return rrt::hfail(RRT_E_INVALID, "null path");
    return rrt::save_png_rgb8(path, rgb8, w, h);
} catch (...) {
    return rrt::host_on_exception();
}

int rrt_host_load_cubemap(rrt_scene* s, const char* folder) try {
    if (!s || !folder) return rrt::hfail(RRT_E_INVALID, "null argument");
    // maps = [left, right, down, up, front, back]   (cubemap.rs:47-48)
    static const char* names[6] = {"left", "right", "down", "up", "front", "back"};
    for (int f = 0; f < 6; ++f) {
        std::vector<uint8_t> px;
        uint32_t w = 0, h = 0;
        const std::string p = std::string(folder) + "/" + names[f] + ".png";
        int rc = rrt::load_png_rgb8(p.c_str(), &px, &w, &h);
        if (rc) return rc;
        rc = rrt_set_sky_face(s, f, px.data(), w, h);
        if (rc) return rc;
    }
    return RRT_OK;
} catch (...) {
    return rrt::host_on_exception();
}

void rrt_host_free(void* p) { std::free(p); }

}  // extern "C"
