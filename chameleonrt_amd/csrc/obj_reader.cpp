// obj_reader.cpp — streaming Wavefront OBJ reader for the harness, behind the small C API of
// include/crt_scene_io.h (libcrt_scene_io.so, plain g++: no HIP in here).
//
// What the reference's importer makes of an OBJ file (util/scene.cpp:94-228, on tinyobjloader): every `o` / `g` group
// that has faces becomes one Geometry; polygons are cut into triangles by tinyobjloader's ear clipping (`triangulate`
// below: concave polygons come out as the reference's triangles, not as a fan); the vertices of a group are re-indexed on
// unique (position, normal, uv) index triples in order of first use; a group's material is the material of its FIRST
// face. This file does the text half of that -- the part that is hopeless line by line in Python on the 10 M-triangle
// assets BASELINE.json names (Rungholt, San Miguel): one pass over the memory-mapped file with std::from_chars
// (correctly rounded, like Python's float() followed by the float32 conversion of the reference loader), an
// open-addressing table per group for the re-indexing. Materials (MTL -> Disney, textures) stay in obj_io.py: a few
// dozen lines of text. chameleonrt_amd/obj_io.py::load_obj uses this reader and keeps its pure-Python twin for the
// tests that demand identical arrays from both (tests/test_obj_io.py).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <charconv>
#include <cmath>
#include <cstddef>
#include <limits>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/crt_scene_io.h"

namespace {

struct Corner {
    int32_t v, n, t;
};

struct Shape {
    std::vector<float> verts, uvs;  // after re-indexing
    std::vector<uint32_t> indices;
    std::string material;           // usemtl name in force at the group's first face ("" = none)
    int material_libs = 0;          // mtllib lines read when that usemtl was met: the name resolves against those only
    bool has_material = false, has_uv = false, mixed_uv = false;
    // re-indexing table: open addressing on the (v, n, t) triple
    std::vector<Corner> keys;
    std::vector<uint32_t> vals;
    size_t used = 0;
    std::vector<Corner> pending;    // corners of the group's faces, three per triangle, before re-indexing
};

inline uint64_t hash_corner(const Corner &c)
{
    uint64_t h = (uint64_t)(uint32_t)c.v * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t)(uint32_t)c.n + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
    h ^= ((uint64_t)(uint32_t)c.t + 0x165667B1ull) * 0x9E3779B185EBCA87ull;
    return h ^ (h >> 29);
}

struct Parser {
    const char *p, *end;
    std::vector<float> pos, tex;
    size_t n_normals = 0;
    std::vector<Shape> shapes;
    std::vector<std::string> mtllibs;
    std::string cur_mat;
    bool have_mat = false;
    int cur_mat_libs = 0;
    int cur = -1; // index into shapes, -1: the next face opens a new one
    std::string error;
    // material ids as tinyobjloader numbers them (every material of every library read so far, in order; a name resolves to the
    // FIRST material that carries it). Only their (in)equality matters here: a `usemtl` that CHANGES the id hands the faces
    // read so far over to the shape, and an `o` statement keeps a shape only if there are faces not handed over yet --
    // see the `o` branch of run().
    std::string base_dir;
    std::vector<std::pair<std::string, int>> material_ids;
    int n_materials = 0, cur_mat_id = -1;
    bool faces_since_change = false;

    static bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\f' || c == '\v'; }
    void skip_space()
    {
        while (p < end && is_space(*p)) {
            ++p;
        }
    }
    const char *token_end() const
    {
        const char *q = p;
        while (q < end && !is_space(*q) && *q != '\n') {
            ++q;
        }
        return q;
    }
    void skip_line()
    {
        while (p < end && *p != '\n') {
            ++p;
        }
        if (p < end) {
            ++p;
        }
    }
    bool at_eol()
    {
        skip_space();
        return p >= end || *p == '\n';
    }
    bool read_float(float &out)
    {
        skip_space();
        const char *q = token_end();
        if (q == p) {
            return false;
        }
        const char *b = p;
        if (*b == '+') {
            ++b; // from_chars does not accept a leading plus; Python's float() does
        }
        // The reference's number grammar (tinyobjloader's tryParseDouble, util/tiny_obj_loader.h:567-680): an optional sign,
        // then at least one DIGIT. ".5", "-.5", "inf", "nan" do not parse there and silently read as 0.0; here such a
        // file is refused rather than loaded as something else than the reference would make of it. (What follows the first
        // digit -- fraction, exponent, the whole token consumed -- is from_chars' general format = the rest of that grammar.)
        {
            const char *d0 = (*b == '-') ? b + 1 : b;
            if (d0 >= q || *d0 < '0' || *d0 > '9') {
                return false;
            }
        }
        double d = 0.0;
        const auto r = std::from_chars(b, q, d);
        if (r.ec != std::errc() || r.ptr != q) {
            return false;
        }
        out = (float)d; // float(x) -> float32, like np.asarray(..., np.float32)
        p = q;
        return true;
    }
    // rest of the line, whitespace-separated tokens joined by single blanks (Python: " ".join(tok[1:]))
    std::string rest_joined()
    {
        std::string s;
        while (!at_eol()) {
            const char *q = token_end();
            if (!s.empty()) {
                s.push_back(' ');
            }
            s.append(p, q);
            p = q;
        }
        return s;
    }
    int material_id_of(const std::string &name) const
    {
        for (const auto &m : material_ids) {
            if (m.first == name) {
                return m.second;
            }
        }
        return -1;
    }
    // The NAMES of a material library, numbered like tinyobjloader's LoadMtl numbers its materials (util/tiny_obj_loader.h:
    // 1353-1725; obj_io.py _parse_mtl is the full statement): a material is flushed by the next `newmtl` only if it has a name,
    // the last one always; the name is what follows `newmtl` and ONE blank, without the line's trailing blanks. A directory
    // ("mtllib  a.mtl": the first name is empty) reads as an empty file. false: the file cannot be opened (the reference throws).
    bool read_material_names(const std::string &file)
    {
        const std::string path = base_dir + file;
        std::string text;
        struct stat st;
        if (stat(path.c_str(), &st) != 0) {
            return false;
        }
        if (!S_ISDIR(st.st_mode)) {
            FILE *f = std::fopen(path.c_str(), "rb");
            if (!f) {
                return false;
            }
            char buf[65536];
            size_t n;
            while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) {
                text.append(buf, n);
            }
            std::fclose(f);
        }
        std::string name;
        auto flush = [&](bool always) {
            if (always || !name.empty()) {
                if (material_id_of(name) < 0) {
                    material_ids.emplace_back(name, n_materials);
                }
                ++n_materials;
            }
        };
        size_t i = 0;
        while (i <= text.size()) {
            size_t e = i;
            while (e < text.size() && text[e] != '\n' && text[e] != '\r') {
                ++e;
            }
            size_t b = i, t = e;
            while (t > b && (text[t - 1] == ' ' || text[t - 1] == '\t')) {
                --t;
            }
            while (b < t && (text[b] == ' ' || text[b] == '\t')) {
                ++b;
            }
            if (t - b > 6 && text.compare(b, 6, "newmtl") == 0 && (text[b + 6] == ' ' || text[b + 6] == '\t')) {
                flush(false);
                name = text.substr(b + 7, t - (b + 7));
            }
            if (e >= text.size()) {
                break;
            }
            i = (text[e] == '\r' && e + 1 < text.size() && text[e + 1] == '\n') ? e + 2 : e + 1;
        }
        flush(true);
        return true;
    }
    // from `from` to the end of the line as it stands (a CR before the LF is not part of the line)
    std::string raw_rest(const char *from) const
    {
        if (from > end) {
            from = end;
        }
        const char *e = from;
        while (e < end && *e != '\n') {
            ++e;
        }
        if (e > from && e[-1] == '\r') {
            --e;
        }
        return std::string(from, e);
    }
    static bool parse_int(const char *b, const char *e, long &out)
    {
        if (b < e && *b == '+') {
            ++b;
        }
        const auto r = std::from_chars(b, e, out);
        return r.ec == std::errc() && r.ptr == e;
    }
    // OBJ indices are 1-based; negative = relative to what has been read so far. Checked in `long`, before narrowing: 0 or an
    // index beyond what has been read (either way) is an error for v and vt -- never a silent "no texcoord". A normal index
    // is never dereferenced (the hot path reads no normals): it only takes part in the (v, vn, vt)
    // re-indexing key, like in the reference's reader (tinyobjloader's fixIndex checks 0 only), so beyond 0 it is only
    // required to fit 32 bits (key_only).
    static bool resolve(long i, size_t n, int32_t &out, bool key_only = false)
    {
        if (i == 0 || (!key_only && (i > (long)n || i < -(long)n)) || i > 0x7fffffffl || i < -0x7fffffffl) {
            return false;
        }
        out = (int32_t)(i > 0 ? i - 1 : (long)n + i);
        return true;
    }

    // ---- polygons -> triangles, as the reference's importer gets them from tinyobjloader (util/tiny_obj_loader.h:1107-1310,
    // called with triangulate = true by util/scene.cpp:117). Not a fan: the polygon is projected on the two axes chosen from
    // its first non-degenerate corner, its signed area gives the winding, and ears are clipped -- starting at corner
    // `guess`, three consecutive remaining corners (a, b, c) are an ear if the turn at b has the polygon's winding and no
    // other remaining corner lies inside (a, b, c) by the crossing-number test; an ear is emitted as (a, b, c) and b is
    // taken out; otherwise the search moves on by one corner. The search gives up after as many fruitless steps as there
    // are corners left (what has been emitted stays); the last three corners are emitted as they are. All arithmetic in
    // float, in the reference's order -- which ears exist depends on it. Consequence worth knowing: the first emitted corner
    // of a face need not be its first corner, so the re-indexed vertex ORDER of a group depends on this, too.
    // (tinyobjloader's "invalid index" skips cannot occur here: parse_face has refused such faces already.)
    static bool inside_tri(const float *vx, const float *vy, float tx, float ty)
    {
        bool c = false;
        for (int i = 0, j = 2; i < 3; j = i++) {
            if (((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) {
                c = !c;
            }
        }
        return c;
    }
    std::vector<Corner> ring; // the corners still to be cut (scratch of triangulate)
    template <class Emit> void triangulate(const std::vector<Corner> &face, Emit &&emit)
    {
        const size_t n = face.size();
        if (n < 3) {
            return;
        }
        if (n == 3) {
            emit(face[0], face[1], face[2]);
            return;
        }
        const float *v = pos.data();
        int ax0 = 1, ax1 = 2;
        for (size_t k = 0; k < n; ++k) {
            const float *a = v + 3 * (size_t)face[k].v, *b = v + 3 * (size_t)face[(k + 1) % n].v, *c = v + 3 * (size_t)face[(k + 2) % n].v;
            const float e0x = b[0] - a[0], e0y = b[1] - a[1], e0z = b[2] - a[2];
            const float e1x = c[0] - b[0], e1y = c[1] - b[1], e1z = c[2] - b[2];
            const float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
            const float eps = std::numeric_limits<float>::epsilon();
            if (cx > eps || cy > eps || cz > eps) { // the first corner that is one
                if (!(cx > cy && cx > cz)) {
                    ax0 = 0;
                    if (cz > cx && cz > cy) {
                        ax1 = 1;
                    }
                }
                break;
            }
        }
        float area = 0.f;
        for (size_t k = 0; k < n; ++k) {
            const float *a = v + 3 * (size_t)face[k].v, *b = v + 3 * (size_t)face[(k + 1) % n].v;
            area += (a[ax0] * b[ax1] - a[ax1] * b[ax0]) * 0.5f;
        }
        ring = face;
        size_t guess = 0, budget = n, last_size = n;
        while (ring.size() > 3 && budget > 0) {
            const size_t m = ring.size();
            if (guess >= m) {
                guess -= m;
            }
            if (last_size != m) { // the previous step cut a corner off: a fresh budget
                last_size = m;
                budget = m;
            } else {
                --budget;
            }
            Corner tri[3];
            float vx[3], vy[3];
            for (int k = 0; k < 3; ++k) {
                tri[k] = ring[(guess + (size_t)k) % m];
                vx[k] = v[3 * (size_t)tri[k].v + ax0];
                vy[k] = v[3 * (size_t)tri[k].v + ax1];
            }
            const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
            const float cross = e0x * e1y - e0y * e1x;
            if (cross * area < 0.f) { // a reflex corner
                ++guess;
                continue;
            }
            bool blocked = false;
            for (size_t o = 3; o < m && !blocked; ++o) {
                const Corner &q = ring[(guess + o) % m];
                blocked = inside_tri(vx, vy, v[3 * (size_t)q.v + ax0], v[3 * (size_t)q.v + ax1]);
            }
            if (blocked) {
                ++guess;
                continue;
            }
            emit(tri[0], tri[1], tri[2]);
            ring.erase(ring.begin() + (std::ptrdiff_t)((guess + 1) % m));
        }
        if (ring.size() == 3) {
            emit(ring[0], ring[1], ring[2]);
        }
    }

    std::vector<Corner> face; // the corners of the face being read (scratch of parse_face)
    bool parse_face()
    {
        face.clear();
        while (!at_eol()) {
            const char *q = token_end();
            const char *s1 = (const char *)memchr(p, '/', (size_t)(q - p));
            const char *s2 = s1 ? (const char *)memchr(s1 + 1, '/', (size_t)(q - s1 - 1)) : nullptr;
            long vi = 0, ti = 0, ni = 0;
            if (!parse_int(p, s1 ? s1 : q, vi)) {
                error = "bad face index";
                return false;
            }
            Corner c;
            c.t = -1;
            c.n = -1;
            if (!resolve(vi, pos.size() / 3, c.v)) {
                error = "face index out of range";
                return false;
            }
            if (s1) {
                const char *te = s2 ? s2 : q;
                if (te > s1 + 1) {
                    if (!parse_int(s1 + 1, te, ti)) {
                        error = "bad face index";
                        return false;
                    }
                    if (!resolve(ti, tex.size() / 2, c.t)) {
                        error = "face index out of range";
                        return false;
                    }
                }
                if (s2 && q > s2 + 1) {
                    if (!parse_int(s2 + 1, q, ni)) {
                        error = "bad face index";
                        return false;
                    }
                    if (!resolve(ni, n_normals, c.n, true)) {
                        error = "face index out of range";
                        return false;
                    }
                }
            }
            p = q;
            face.push_back(c);
        }
        triangulate(face, [this](const Corner &a, const Corner &b, const Corner &c) {
            if (cur < 0) {
                shapes.emplace_back();
                cur = (int)shapes.size() - 1;
            }
            Shape &s = shapes[(size_t)cur];
            if (s.pending.empty()) {
                s.material = cur_mat;
                s.has_material = have_mat;
                s.material_libs = cur_mat_libs;
            }
            s.pending.push_back(a);
            s.pending.push_back(b);
            s.pending.push_back(c);
        });
        return true;
    }

    bool run()
    {
        while (p < end) {
            skip_space();
            if (p >= end) {
                break;
            }
            if (*p == '\n' || *p == '#') {
                skip_line();
                continue;
            }
            const char *q = token_end();
            size_t len = (size_t)(q - p);
            const char *kw = p;
            p = q;
            // tinyobjloader recognises a statement by its keyword AND a blank (space or tab) right after it: `o`, `g`, `usemtl`, `v` ...
            // alone on a line are not statements at all and are skipped (util/tiny_obj_loader.h: `token[0] == 'g' && IS_SPACE(token[1])`)
            if (!(q < end && (*q == ' ' || *q == '\t'))) {
                len = 0;
            }
            if (len == 1 && kw[0] == 'v') {
                float x[3];
                if (!read_float(x[0]) || !read_float(x[1]) || !read_float(x[2])) {
                    error = "bad vertex";
                    return false;
                }
                pos.insert(pos.end(), x, x + 3);
            } else if (len == 2 && kw[0] == 'v' && kw[1] == 't') {
                float u = 0.f, v = 0.f;
                if (!read_float(u)) {
                    error = "bad texture coordinate";
                    return false;
                }
                if (!at_eol() && !read_float(v)) {
                    error = "bad texture coordinate";
                    return false;
                }
                tex.push_back(u);
                tex.push_back(v);
            } else if (len == 2 && kw[0] == 'v' && kw[1] == 'n') {
                ++n_normals; // vertex normals only take part in the re-indexing key (the renderer ignores them: quirk Q7)
            } else if (len == 1 && kw[0] == 'f') {
                faces_since_change = true;
                if (!parse_face()) {
                    return false;
                }
            } else if (len == 1 && (kw[0] == 'o' || kw[0] == 'g')) {
                // tinyobjloader hands the faces read so far to the current shape when the material changes, and its `o` statement
                // keeps the shape only if there are faces it has not handed over yet (util/tiny_obj_loader.h:2117-2123; `g` and
                // the end of the file look at the shape itself): an `o` right after a material-changing `usemtl` LOSES the
                // object before it. The reference's scenes are what that makes of a file, so the faces are dropped here, too.
                if (kw[0] == 'o' && cur >= 0 && !shapes[(size_t)cur].pending.empty() && !faces_since_change) {
                    shapes[(size_t)cur].pending.clear();
                }
                faces_since_change = false;
                // a new group; tinyobj does not emit empty shapes, so one without faces is simply continued
                if (!(cur >= 0 && shapes[(size_t)cur].pending.empty())) {
                    cur = -1;
                }
            } else if (len == 6 && std::memcmp(kw, "usemtl", 6) == 0) {
                // the name is everything after the keyword and ONE blank, to the end of the line, as it stands (tinyobjloader:
                // `token += 7; ss << token`): "usemtl  a" names " a", "usemtl a " names "a " -- neither is the material "a"
                cur_mat = raw_rest(p + 1);
                have_mat = true;
                cur_mat_libs = (int)mtllibs.size();
                const int id = material_id_of(cur_mat);
                if (id != cur_mat_id) {
                    cur_mat_id = id;
                    faces_since_change = false;
                }
            } else if (len == 6 && std::memcmp(kw, "mtllib", 6) == 0) {
                // File names separated by single blanks (SplitString = std::getline with ' ', util/tiny_obj_loader.h:1343-1351: two
                // blanks in a row name "" in between, a trailing blank names nothing). tinyobjloader tries them in turn and stops at
                // the first that OPENS (tiny_obj_loader.h:2031-2049; MaterialFileReader, :1741-1749, only WARNS about a file it
                // cannot open and returns false). If none opens it warns "Failed to load material file(s). Use default material."
                // and goes on: no material is defined by this statement, every later `usemtl` of an unknown name resolves to -1,
                // and the reference's importer gives those geometries its default material (Scene::load_obj throws only on
                // !ret || !err.empty(), util/scene.cpp:110; a missing .mtl sets neither). OBJ files whose .mtl is absent are common.
                const std::string rest = raw_rest(p + 1);
                bool found = false;
                size_t at = 0;
                while (!found && at < rest.size()) {
                    const size_t blank = rest.find(' ', at);
                    const std::string name = rest.substr(at, blank == std::string::npos ? std::string::npos : blank - at);
                    if (read_material_names(name)) {
                        mtllibs.emplace_back(name);
                        found = true;
                    }
                    if (blank == std::string::npos) {
                        break;
                    }
                    at = blank + 1;
                }
                if (!found) {
                    std::fprintf(stderr, "[crt_scene_io] no material library of `mtllib %s` could be opened in %s: the default material is used\n",
                                 rest.c_str(), base_dir.empty() ? "." : base_dir.c_str());
                }
            }
            skip_line();
        }
        // re-index every group on unique (v, n, t) triples in order of first use
        for (Shape &s : shapes) {
            size_t cap = 16;
            while (cap < 2 * s.pending.size()) {
                cap *= 2;
            }
            s.keys.assign(cap, Corner{-2, -2, -2});
            s.vals.assign(cap, 0u);
            s.indices.reserve(s.pending.size());
            size_t n_uv = 0;
            for (const Corner &c : s.pending) {
                size_t h = (size_t)hash_corner(c) & (cap - 1);
                for (;;) {
                    const Corner &k = s.keys[h];
                    if (k.v == -2) {
                        s.keys[h] = c;
                        s.vals[h] = (uint32_t)(s.verts.size() / 3);
                        s.verts.insert(s.verts.end(), pos.begin() + 3 * (size_t)c.v, pos.begin() + 3 * (size_t)c.v + 3);
                        if (c.t >= 0) {
                            s.uvs.insert(s.uvs.end(), tex.begin() + 2 * (size_t)c.t, tex.begin() + 2 * (size_t)c.t + 2);
                            ++n_uv;
                        }
                        break;
                    }
                    if (k.v == c.v && k.n == c.n && k.t == c.t) {
                        break;
                    }
                    h = (h + 1) & (cap - 1);
                }
                s.indices.push_back(s.vals[h]);
            }
            s.has_uv = n_uv > 0;
            s.mixed_uv = n_uv > 0 && n_uv != s.verts.size() / 3;
            s.keys = std::vector<Corner>();
            s.vals = std::vector<uint32_t>();
            s.pending = std::vector<Corner>();
        }
        // groups without faces never became shapes
        return true;
    }
};

} // namespace

struct crt_obj_file {
    Parser parser;
    std::string error;
};

extern "C" {

crt_obj_file *crt_obj_parse(const char *path)
{
    crt_obj_file *f = new crt_obj_file;
    const int fd = path ? open(path, O_RDONLY) : -1;
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) {
        f->error = std::string("cannot read ") + (path ? path : "(null)");
        if (fd >= 0) {
            close(fd);
        }
        return f;
    }
    const size_t size = (size_t)st.st_size;
    void *map = size ? mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    close(fd);
    if (size && map == MAP_FAILED) {
        f->error = std::string("cannot map ") + path;
        return f;
    }
    {
        const std::string sp(path);
        const size_t slash = sp.rfind('/');
        f->parser.base_dir = slash == std::string::npos ? std::string() : sp.substr(0, slash + 1);
    }
    f->parser.p = static_cast<const char *>(map);
    f->parser.end = f->parser.p + size;
    if (!f->parser.run()) {
        f->error = f->parser.error.empty() ? "parse error" : f->parser.error;
    }
    if (map) {
        munmap(map, size);
    }
    f->parser.pos = std::vector<float>();
    f->parser.tex = std::vector<float>();
    return f;
}

const char *crt_obj_error(const crt_obj_file *f) { return f && !f->error.empty() ? f->error.c_str() : nullptr; }
void crt_obj_free(crt_obj_file *f) { delete f; }
int crt_obj_num_shapes(const crt_obj_file *f) { return f ? (int)f->parser.shapes.size() : 0; }
int crt_obj_num_mtllibs(const crt_obj_file *f) { return f ? (int)f->parser.mtllibs.size() : 0; }
const char *crt_obj_mtllib(const crt_obj_file *f, int i) { return f->parser.mtllibs[(size_t)i].c_str(); }

int crt_obj_shape_info(const crt_obj_file *f, int s, uint64_t *n_vertices, uint64_t *n_triangles, int *has_uv, int *has_material)
{
    if (!f || s < 0 || (size_t)s >= f->parser.shapes.size()) {
        return -1;
    }
    const Shape &sh = f->parser.shapes[(size_t)s];
    *n_vertices = sh.verts.size() / 3;
    *n_triangles = sh.indices.size() / 3;
    *has_uv = sh.mixed_uv ? -1 : (sh.has_uv ? 1 : 0); // -1: the group mixes vertices with and without texture coordinates
    *has_material = sh.has_material ? 1 : 0;
    return 0;
}
const char *crt_obj_shape_material(const crt_obj_file *f, int s) { return f->parser.shapes[(size_t)s].material.c_str(); }
int crt_obj_shape_material_libs(const crt_obj_file *f, int s) { return f->parser.shapes[(size_t)s].material_libs; }

int crt_obj_shape_copy(const crt_obj_file *f, int s, float *vertices, uint32_t *indices, float *uvs)
{
    if (!f || s < 0 || (size_t)s >= f->parser.shapes.size()) {
        return -1;
    }
    const Shape &sh = f->parser.shapes[(size_t)s];
    std::memcpy(vertices, sh.verts.data(), sh.verts.size() * sizeof(float));
    std::memcpy(indices, sh.indices.data(), sh.indices.size() * sizeof(uint32_t));
    if (uvs && sh.has_uv && !sh.mixed_uv) {
        std::memcpy(uvs, sh.uvs.data(), sh.uvs.size() * sizeof(float));
    }
    return 0;
}

} // extern "C"
