// host_scene.cpp — host plumbing in front of the device core: scene text ->
// rt_scene_desc.  The reference keeps this side in Rust (src/sdl.rs,
// src/sdl_grammar.rs, wavefront_obj, image); no Rust toolchain exists in this
// image, so the same grammar is written here as a recursive-descent parser.
// It runs once per scene and is not part of the accelerated path.
//
// Grammar: reference src/sdl_grammar.rs:19-324 (PEG, ordered choice, no
// keyword boundaries).  Deviations, all on the permissive side:
//   * `homogenous_medium` and `mesh` accept whitespace/comments between their
//     fields (the reference rules at sdl_grammar.rs:108-111,126-129 lack `_`
//     there, which makes scenes/fog_light.scene unparseable upstream).
//   * images: JPEG decoding (image 0.14 / jpeg-decoder) is not reproduced; an
//     `image "x.jpg"` texture loads the binary PPM `x.ppm` next to it.
#include "../../include/rt_api.h"
#include "host_jpeg.h"
#include "host_math.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace rt_host {

void set_last_error(const std::string &msg); // rt_capi_common.cpp

struct SceneOwner {
    rt_scene_desc desc; // must stay the first member
    std::vector<rt_object> objects;
    std::vector<rt_shape_node> nodes;
    std::vector<int32_t> child_list;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_image> images;
    std::vector<uint8_t> image_data;
    std::vector<rt_mesh> meshes;
    std::vector<double> vertices, normals;
    std::vector<uint32_t> tri_v, tri_n;

    void finalize()
    {
        desc.n_objects = (uint32_t)objects.size();
        desc.objects = objects.data();
        desc.n_nodes = (uint32_t)nodes.size();
        desc.nodes = nodes.data();
        desc.n_child_list = (uint32_t)child_list.size();
        desc.child_list = child_list.data();
        desc.n_materials = (uint32_t)materials.size();
        desc.materials = materials.data();
        desc.n_textures = (uint32_t)textures.size();
        desc.textures = textures.data();
        desc.n_images = (uint32_t)images.size();
        desc.images = images.data();
        desc.n_image_bytes = image_data.size();
        desc.image_data = image_data.data();
        desc.n_meshes = (uint32_t)meshes.size();
        desc.meshes = meshes.data();
        desc.n_vertices = (uint32_t)(vertices.size() / 3);
        desc.vertices = vertices.data();
        desc.n_normals = (uint32_t)(normals.size() / 3);
        desc.normals = normals.data();
        desc.n_triangles = (uint32_t)(tri_v.size() / 3);
        desc.tri_v = tri_v.data();
        desc.tri_n = tri_n.data();
    }
};

struct ParseError {
    int status;
    std::string msg;
};

static std::string join_path(const std::string &base, const std::string &p)
{
    if (p.empty() || p[0] == '/' || base.empty()) return p;
    if (base.back() == '/') return base + p;
    return base + "/" + p;
}

// ---- wavefront OBJ -------------------------------------------------------
// sdl.rs:55-92 via wavefront_obj 5.1.0 (not vendored in the reference; its
// behaviour is restated from its documentation: per-`o` objects with their own
// 0-based vertex/normal pools, polygons fan-triangulated as (v0, v_i, v_i+1),
// only faces with normals are usable because sdl.rs:79 unwraps them).
static void load_obj(SceneOwner &sc, const std::string &path, std::vector<int32_t> &mesh_nodes)
{
    std::ifstream f(path);
    if (!f) throw ParseError{RT_ERR_IO, "could not open object file: " + path};

    struct Obj {
        std::vector<double> v, n;
        std::vector<uint32_t> tv, tn;
    };
    std::vector<Obj> objs;
    size_t v_base = 0, n_base = 0; // global 1-based -> per-object 0-based
    size_t v_total = 0, n_total = 0;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag)) continue;
        if (tag == "o") {
            objs.emplace_back();
            v_base = v_total;
            n_base = n_total;
        } else if (tag == "v" || tag == "vn") {
            if (objs.empty()) objs.emplace_back();
            double x, y, z;
            if (!(ls >> x >> y >> z)) throw ParseError{RT_ERR_PARSE, "Could not parse object file: " + line};
            std::vector<double> &dst = (tag == "v") ? objs.back().v : objs.back().n;
            dst.push_back(x);
            dst.push_back(y);
            dst.push_back(z);
            if (tag == "v")
                ++v_total;
            else
                ++n_total;
        } else if (tag == "f") {
            if (objs.empty()) objs.emplace_back();
            std::vector<long> vi, ni;
            std::string tok;
            while (ls >> tok) {
                long a = 0, c = 0;
                bool has_n = false;
                size_t s1 = tok.find('/');
                a = std::strtol(tok.substr(0, s1).c_str(), nullptr, 10);
                if (s1 != std::string::npos) {
                    size_t s2 = tok.find('/', s1 + 1);
                    if (s2 != std::string::npos && s2 + 1 < tok.size()) {
                        c = std::strtol(tok.substr(s2 + 1).c_str(), nullptr, 10);
                        has_n = true;
                    }
                }
                if (a < 0) a = (long)v_total + 1 + a;
                if (has_n && c < 0) c = (long)n_total + 1 + c;
                if (!has_n) throw ParseError{RT_ERR_PARSE, "mesh face without normals (sdl.rs:79 unwrap): " + line};
                vi.push_back(a - 1 - (long)v_base);
                ni.push_back(c - 1 - (long)n_base);
            }
            if (vi.size() < 3) continue; // points and lines are not Primitive::Triangle (sdl.rs:81)
            Obj &o = objs.back();
            for (size_t i = 1; i + 1 < vi.size(); ++i) {
                long idx[3] = {vi[0], vi[i], vi[i + 1]};
                long nidx[3] = {ni[0], ni[i], ni[i + 1]};
                for (int k = 0; k < 3; ++k) {
                    if (idx[k] < 0 || (size_t)idx[k] >= o.v.size() / 3 || nidx[k] < 0 ||
                        (size_t)nidx[k] >= o.n.size() / 3)
                        throw ParseError{RT_ERR_PARSE, "mesh index out of range: " + line};
                    o.tv.push_back((uint32_t)idx[k]);
                    o.tn.push_back((uint32_t)nidx[k]);
                }
            }
        }
    }
    for (const Obj &o : objs) {
        rt_mesh m;
        m.first_vertex = (uint32_t)(sc.vertices.size() / 3);
        m.n_vertices = (uint32_t)(o.v.size() / 3);
        m.first_normal = (uint32_t)(sc.normals.size() / 3);
        m.n_normals = (uint32_t)(o.n.size() / 3);
        m.first_triangle = (uint32_t)(sc.tri_v.size() / 3);
        m.n_triangles = (uint32_t)(o.tv.size() / 3);
        sc.vertices.insert(sc.vertices.end(), o.v.begin(), o.v.end());
        sc.normals.insert(sc.normals.end(), o.n.begin(), o.n.end());
        sc.tri_v.insert(sc.tri_v.end(), o.tv.begin(), o.tv.end());
        sc.tri_n.insert(sc.tri_n.end(), o.tn.begin(), o.tn.end());
        sc.meshes.push_back(m);

        rt_shape_node node;
        std::memset(&node, 0, sizeof(node));
        node.kind = RT_SHAPE_MESH;
        node.child_a = (int32_t)sc.meshes.size() - 1;
        node.child_b = -1;
        node.flags = 1; // Mesh::new(..., true): smooth shading (sdl.rs:85)
        mat_store(mat_identity(), node.o2w);
        mat_store(mat_identity(), node.w2o);
        sc.nodes.push_back(node);
        mesh_nodes.push_back((int32_t)sc.nodes.size() - 1);
    }
}

// ---- images ----------------------------------------------------------------
static int load_image(SceneOwner &sc, const std::string &path)
{
    std::string p = path;
    size_t dot = p.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : p.substr(dot);
    for (char &c : ext) c = (char)std::tolower((unsigned char)c);
    if (ext == ".jpg" || ext == ".jpeg") {
        // image::open (sdl.rs:52): decoded here by host_jpeg.cpp.  A file that is absent falls through to a
        // decoded PPM of the same name, which is how the fixtures pin the texel bytes of earth.jpg.
        std::ifstream jf(p, std::ios::binary);
        if (jf) {
            std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(jf)), std::istreambuf_iterator<char>());
            int w = 0, h = 0;
            std::vector<uint8_t> rgb;
            std::string err;
            if (!decode_jpeg(bytes.data(), bytes.size(), w, h, rgb, err))
                throw ParseError{RT_ERR_IO, "could not decode image file: " + p + ": " + err};
            rt_image im;
            im.width = (uint32_t)w;
            im.height = (uint32_t)h;
            im.offset = sc.image_data.size();
            sc.image_data.insert(sc.image_data.end(), rgb.begin(), rgb.end());
            sc.images.push_back(im);
            return (int)sc.images.size() - 1;
        }
    }
    if (ext != ".ppm") p = (dot == std::string::npos ? p : p.substr(0, dot)) + ".ppm";
    std::ifstream f(p, std::ios::binary);
    if (!f) throw ParseError{RT_ERR_IO, "could not open image file: " + p + " (decoded PPM stand-in for " + path + ")"};
    std::string magic;
    f >> magic;
    if (magic != "P6") throw ParseError{RT_ERR_IO, "could not decode image file: " + p};
    auto next_int = [&]() -> long {
        for (;;) {
            int c = f.peek();
            if (c == '#') {
                std::string skip;
                std::getline(f, skip);
            } else if (std::isspace(c)) {
                f.get();
            } else
                break;
        }
        long v = -1;
        f >> v;
        return v;
    };
    long w = next_int(), h = next_int(), maxv = next_int();
    f.get(); // single whitespace after maxval
    if (w <= 0 || h <= 0 || maxv != 255) throw ParseError{RT_ERR_IO, "could not decode image file: " + p};
    rt_image im;
    im.width = (uint32_t)w;
    im.height = (uint32_t)h;
    im.offset = sc.image_data.size();
    size_t bytes = (size_t)w * (size_t)h * 3;
    sc.image_data.resize(im.offset + bytes);
    f.read((char *)sc.image_data.data() + im.offset, (std::streamsize)bytes);
    if ((size_t)f.gcount() != bytes) throw ParseError{RT_ERR_IO, "could not decode image file: " + p};
    sc.images.push_back(im);
    return (int)sc.images.size() - 1;
}

// ---- the parser ------------------------------------------------------------
class Parser {
  public:
    Parser(const char *text, const std::string &base_dir, uint32_t width, uint32_t height, SceneOwner &sc)
        : s_(text), n_(std::strlen(text)), base_(base_dir), width_(width), height_(height), sc_(sc)
    {
    }

    // sdl_grammar.rs:19-26
    void scene()
    {
        std::memset(&sc_.desc, 0, sizeof(sc_.desc));
        double bg[3] = {0.0, 0.0, 0.0}; // SceneOptions::default (sdl.rs:28-34)
        options(bg);
        ws();
        if (!camera()) fail("camera");
        ws();
        int count = 0;
        for (;;) {
            size_t save = pos_;
            if (!object()) {
                pos_ = save;
                break;
            }
            ws();
            ++count;
        }
        if (count == 0) fail("object");
        if (pos_ != n_) fail("object or end of input");
        for (int k = 0; k < 3; ++k) sc_.desc.background[k] = bg[k];
    }

  private:
    const char *s_;
    size_t n_;
    size_t pos_ = 0;
    size_t far_ = 0;
    std::string far_what_;
    std::string base_;
    uint32_t width_, height_;
    SceneOwner &sc_;

    [[noreturn]] void fail(const char *what)
    {
        note(what);
        size_t line = 1, col = 1;
        for (size_t i = 0; i < far_ && i < n_; ++i) {
            if (s_[i] == '\n') {
                ++line;
                col = 1;
            } else
                ++col;
        }
        std::ostringstream os;
        os << "could not parse scene file: error at " << line << ":" << col << ": expected " << far_what_;
        throw ParseError{RT_ERR_PARSE, os.str()};
    }
    void note(const char *what)
    {
        if (pos_ >= far_) {
            if (pos_ > far_) far_what_.clear();
            far_ = pos_;
            if (!far_what_.empty()) far_what_ += ", ";
            far_what_ += what;
        }
    }

    bool lit(const char *k)
    {
        size_t len = std::strlen(k);
        if (pos_ + len <= n_ && std::memcmp(s_ + pos_, k, len) == 0) {
            pos_ += len;
            return true;
        }
        note(k);
        return false;
    }
    static bool is_ws(char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }

    // rule _ (sdl_grammar.rs:320-324)
    void ws()
    {
        for (;;) {
            if (pos_ < n_ && is_ws(s_[pos_])) {
                ++pos_;
            } else if (pos_ + 1 < n_ && s_[pos_] == '/' && s_[pos_ + 1] == '/') {
                pos_ += 2;
                while (pos_ < n_ && s_[pos_] != '\n') ++pos_;
            } else if (pos_ + 1 < n_ && s_[pos_] == '/' && s_[pos_ + 1] == '*') {
                // "/*" [^'*']* "*/": a '*' inside the body ends the match attempt
                size_t p = pos_ + 2;
                while (p < n_ && s_[p] != '*') ++p;
                if (p + 1 < n_ && s_[p] == '*' && s_[p + 1] == '/')
                    pos_ = p + 2;
                else
                    return;
            } else
                return;
        }
    }

    // rule float (sdl_grammar.rs:307-313)
    bool number(double &out)
    {
        size_t p = pos_;
        if (p < n_ && s_[p] == '-') ++p;
        size_t d0 = p;
        while (p < n_ && std::isdigit((unsigned char)s_[p])) ++p;
        if (p == d0) {
            note("float literal");
            return false;
        }
        if (p + 1 < n_ && s_[p] == '.' && std::isdigit((unsigned char)s_[p + 1])) {
            size_t q = p + 1;
            while (q < n_ && std::isdigit((unsigned char)s_[q])) ++q;
            p = q;
            if (p + 1 < n_ && (s_[p] == 'e' || s_[p] == 'E') && std::isdigit((unsigned char)s_[p + 1])) {
                q = p + 1;
                while (q < n_ && std::isdigit((unsigned char)s_[q])) ++q;
                p = q;
            }
        }
        std::string tok(s_ + pos_, p - pos_);
        out = std::strtod(tok.c_str(), nullptr);
        pos_ = p;
        return true;
    }

    // rule string (sdl_grammar.rs:261-264)
    bool string_lit(std::string &out)
    {
        size_t save = pos_;
        if (!lit("\"")) return false;
        size_t p = pos_;
        while (p < n_ && s_[p] != '"') ++p;
        if (p >= n_) {
            pos_ = save;
            note("\"");
            return false;
        }
        out.assign(s_ + pos_, p - pos_);
        pos_ = p + 1;
        while (pos_ < n_ && is_ws(s_[pos_])) ++pos_;
        return true;
    }

    // rules vec3 / full_vec3 / short_vec3 (sdl_grammar.rs:297-305)
    bool vec3(double v[3])
    {
        size_t save = pos_;
        if (lit("<")) {
            ws();
            double a, b, c;
            if (number(a)) {
                ws();
                size_t mid = pos_;
                if (lit(",")) {
                    ws();
                    if (number(b)) {
                        ws();
                        if (lit(",")) {
                            ws();
                            if (number(c)) {
                                ws();
                                if (lit(">")) {
                                    v[0] = a;
                                    v[1] = b;
                                    v[2] = c;
                                    return true;
                                }
                            }
                        }
                    }
                }
                pos_ = mid;
                if (lit(">")) {
                    v[0] = v[1] = v[2] = a;
                    return true;
                }
            }
        }
        pos_ = save;
        return false;
    }

    // rule direction (sdl_grammar.rs:271-278)
    bool direction(double v[3])
    {
        if (vec3(v)) return true;
        struct {
            const char *k;
            double x, y, z;
        } named[] = {{"down", 0, -1, 0}, {"up", 0, 1, 0},    {"left", -1, 0, 0},
                     {"right", 1, 0, 0}, {"back", 0, 0, -1}, {"front", 0, 0, 1}};
        for (auto &nm : named)
            if (lit(nm.k)) {
                v[0] = nm.x;
                v[1] = nm.y;
                v[2] = nm.z;
                return true;
            }
        return false;
    }

    // rules color / std_color / rgb_color (sdl_grammar.rs:280-295)
    bool color(double c[3])
    {
        size_t save = pos_;
        if (!(lit("color") || lit("colour"))) return false;
        ws();
        struct {
            const char *k;
            double r, g, b;
        } named[] = {{"white", 1, 1, 1},  {"black", 0, 0, 0},  {"red", 1, 0, 0},  {"green", 0, 1, 0},
                     {"blue", 0, 0, 1},   {"yellow", 1, 1, 0}, {"cyan", 0, 1, 1}, {"magenta", 1, 0, 1}};
        for (auto &nm : named)
            if (lit(nm.k)) {
                c[0] = nm.r;
                c[1] = nm.g;
                c[2] = nm.b;
                return true;
            }
        if (lit("rgb")) {
            ws();
            if (vec3(c)) return true;
        }
        pos_ = save;
        return false;
    }

    // keyword _ float helpers (radius, width, height, density, fuzz, ior, intensity, fov)
    bool keyed_float(const char *key, double &out)
    {
        size_t save = pos_;
        if (lit(key)) {
            ws();
            if (number(out)) return true;
        }
        pos_ = save;
        return false;
    }
    bool keyed_point(const char *key, double v[3])
    {
        size_t save = pos_;
        if (lit(key)) {
            ws();
            if (vec3(v)) return true;
        }
        pos_ = save;
        return false;
    }

    // sdl_grammar.rs:28-35; only tried at the very start of the text
    void options(double bg[3])
    {
        size_t save = pos_;
        if (lit("options")) {
            ws();
            if (lit("{")) {
                ws();
                size_t s2 = pos_;
                if (lit("background")) {
                    ws();
                    if (color(bg)) {
                        ws();
                        if (lit("}")) return;
                    }
                }
                (void)s2;
            }
        }
        pos_ = save;
    }

    // sdl_grammar.rs:37-44, Camera::new system.rs:37-55
    bool camera()
    {
        size_t save = pos_;
        double o[3], p[3], fov = 60.0;
        if (lit("camera")) {
            ws();
            if (lit("{")) {
                ws();
                if (keyed_point("origin", o)) {
                    ws();
                    if (keyed_point("look_at", p)) {
                        ws();
                        double f;
                        if (keyed_float("fov", f)) fov = f;
                        ws();
                        if (lit("}")) {
                            make_camera(o, p, fov);
                            return true;
                        }
                    }
                }
            }
        }
        pos_ = save;
        return false;
    }

    static void normalize3(double v[3])
    {
        // direction.rs:57-65
        double l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
        if (l2 > 0.0) {
            double inv = 1.0 / std::sqrt(l2);
            v[0] *= inv;
            v[1] *= inv;
            v[2] *= inv;
        }
    }
    static void cross3(const double a[3], const double b[3], double r[3])
    {
        // direction.rs:41-47
        r[0] = a[1] * b[2] - a[2] * b[1];
        r[1] = a[2] * b[0] - a[0] * b[2];
        r[2] = a[0] * b[1] - a[1] * b[0];
    }

    void make_camera(const double origin[3], const double look_at[3], double fov)
    {
        double up[3] = {0.0, 1.0, 0.0};
        double z[3] = {origin[0] - look_at[0], origin[1] - look_at[1], origin[2] - look_at[2]};
        normalize3(z);
        normalize3(up);
        double x[3], y[3];
        cross3(up, z, x); // not re-normalised
        cross3(z, x, y);
        double *m = sc_.desc.camera.camera_to_world;
        double rows[4][4] = {{x[0], x[1], x[2], 0.0},
                             {y[0], y[1], y[2], 0.0},
                             {z[0], z[1], z[2], 0.0},
                             {origin[0], origin[1], origin[2], 1.0}};
        std::memcpy(m, rows, sizeof(rows));
        sc_.desc.camera.fov_factor = std::tan(deg_to_rad(fov * 0.5));
        (void)width_;
        (void)height_;
    }

    // sdl_grammar.rs:46-49
    bool object()
    {
        size_t save = pos_;
        if (lit("object")) {
            ws();
            std::string name;
            string_lit(name);
            ws();
            if (lit("{")) {
                ws();
                int shape = object_shape();
                if (shape >= 0) {
                    ws();
                    int mat = object_material();
                    if (mat >= 0) {
                        ws();
                        if (lit("}")) {
                            rt_object o;
                            o.shape = shape;
                            o.material = mat;
                            sc_.objects.push_back(o);
                            return true;
                        }
                    }
                }
            }
        }
        pos_ = save;
        return false;
    }

    // sdl_grammar.rs:51-54
    int object_shape()
    {
        int s = planar_shape();
        if (s >= 0) return s;
        s = solid_shape();
        if (s >= 0) return s;
        return mesh();
    }

    int new_node(int kind)
    {
        rt_shape_node node;
        std::memset(&node, 0, sizeof(node));
        node.kind = kind;
        node.child_a = node.child_b = -1;
        mat_store(mat_identity(), node.o2w);
        mat_store(mat_identity(), node.w2o);
        sc_.nodes.push_back(node);
        return (int)sc_.nodes.size() - 1;
    }

    // sdl::transform_shape (sdl.rs:44-47) -> Transformation::transform (object.rs:21-26)
    void apply_transform(int node, bool has, const Mat4 &m)
    {
        Mat4 o2w = mat_mul(mat_load(sc_.nodes[node].o2w), has ? m : mat_identity());
        Mat4 w2o = mat_inverse(o2w);
        mat_store(o2w, sc_.nodes[node].o2w);
        mat_store(w2o, sc_.nodes[node].w2o);
    }

    // rule transforms (sdl_grammar.rs:196-234), combine_transforms sdl.rs:94-96
    bool transforms(Mat4 &out)
    {
        size_t save = pos_;
        if (lit("transform")) {
            ws();
            if (lit("{")) {
                ws();
                Mat4 acc = mat_identity();
                for (;;) {
                    size_t s2 = pos_;
                    Mat4 m;
                    double v[3], a;
                    bool ok = false;
                    if (lit("translate")) {
                        ws();
                        if (direction(v)) {
                            m = mat_translation(v[0], v[1], v[2]);
                            ok = true;
                        }
                    }
                    if (!ok) {
                        pos_ = s2;
                        if (keyed_float("rotate_x", a)) {
                            m = mat_rotation_x(a);
                            ok = true;
                        } else if (keyed_float("rotate_y", a)) {
                            m = mat_rotation_y(a);
                            ok = true;
                        } else if (keyed_float("rotate_z", a)) {
                            m = mat_rotation_z(a);
                            ok = true;
                        }
                    }
                    if (!ok) {
                        pos_ = s2;
                        if (lit("scale")) {
                            ws();
                            if (direction(v)) {
                                m = mat_scaling(v[0], v[1], v[2]);
                                ok = true;
                            }
                        }
                    }
                    if (!ok) {
                        pos_ = s2;
                        break;
                    }
                    acc = mat_mul(acc, m);
                    ws();
                }
                ws();
                if (lit("}")) {
                    out = acc;
                    return true;
                }
            }
        }
        pos_ = save;
        return false;
    }

    // tail shared by every shape rule: transforms()? _ "}"
    bool finish_shape(int node)
    {
        Mat4 m;
        bool has = transforms(m);
        ws();
        if (!lit("}")) return false;
        apply_transform(node, has, m);
        return true;
    }

    void drop_nodes(size_t n_nodes, size_t n_children, size_t n_meshes)
    {
        // a failed alternative must not leave nodes behind
        sc_.nodes.resize(n_nodes);
        sc_.child_list.resize(n_children);
        if (sc_.meshes.size() > n_meshes) {
            const rt_mesh &m = sc_.meshes[n_meshes];
            sc_.vertices.resize((size_t)m.first_vertex * 3);
            sc_.normals.resize((size_t)m.first_normal * 3);
            sc_.tri_v.resize((size_t)m.first_triangle * 3);
            sc_.tri_n.resize((size_t)m.first_triangle * 3);
            sc_.meshes.resize(n_meshes);
        }
    }

    // sdl_grammar.rs:113-124,133-146
    int planar_shape()
    {
        size_t save = pos_;
        size_t nn = sc_.nodes.size();
        // plane
        if (lit("plane")) {
            ws();
            if (lit("{")) {
                ws();
                double o[3] = {0, 0, 0}, nrm[3];
                keyed_point("origin", o);
                ws();
                size_t s2 = pos_;
                bool ok = false;
                if (lit("normal")) {
                    ws();
                    ok = direction(nrm);
                }
                if (ok) {
                    ws();
                    int node = new_node(RT_SHAPE_PLANE);
                    double *p = sc_.nodes[node].params;
                    p[0] = o[0];
                    p[1] = o[1];
                    p[2] = o[2];
                    p[3] = nrm[0];
                    p[4] = nrm[1];
                    p[5] = nrm[2];
                    if (finish_shape(node)) return node;
                }
                (void)s2;
            }
        }
        pos_ = save;
        sc_.nodes.resize(nn);
        const char *names[3] = {"xyrect", "xzrect", "zyrect"};
        const int kinds[3] = {RT_SHAPE_XYRECT, RT_SHAPE_XZRECT, RT_SHAPE_ZYRECT};
        for (int k = 0; k < 3; ++k) {
            pos_ = save;
            if (!lit(names[k])) continue;
            ws();
            if (!lit("{")) continue;
            ws();
            double o[3] = {0, 0, 0}, w, h;
            keyed_point("origin", o);
            ws();
            if (!keyed_float("width", w)) continue;
            ws();
            if (!keyed_float("height", h)) continue;
            ws();
            bool rev = lit("reverse");
            ws();
            int node = new_node(kinds[k]);
            double *p = sc_.nodes[node].params;
            p[0] = o[0];
            p[1] = o[1];
            p[2] = o[2];
            p[3] = w;
            p[4] = h;
            sc_.nodes[node].flags = rev ? 1 : 0;
            if (finish_shape(node)) return node;
            sc_.nodes.resize(nn);
        }
        pos_ = save;
        return -1;
    }

    // sdl_grammar.rs:58-111
    int solid_shape()
    {
        size_t save = pos_;
        size_t nn = sc_.nodes.size(), nc = sc_.child_list.size(), nm = sc_.meshes.size();
        auto reset = [&]() {
            pos_ = save;
            drop_nodes(nn, nc, nm);
        };
        // sphere
        if (lit("sphere")) {
            ws();
            if (lit("{")) {
                ws();
                double o[3] = {0, 0, 0}, r = 1.0;
                keyed_point("origin", o);
                ws();
                keyed_float("radius", r);
                ws();
                int node = new_node(RT_SHAPE_SPHERE);
                double *p = sc_.nodes[node].params;
                p[0] = o[0];
                p[1] = o[1];
                p[2] = o[2];
                p[3] = r;
                if (finish_shape(node)) return node;
            }
        }
        reset();
        if (lit("cylinder")) {
            ws();
            if (lit("{")) {
                ws();
                double r = 1.0, h = 1.0;
                keyed_float("radius", r);
                ws();
                keyed_float("height", h);
                ws();
                int node = new_node(RT_SHAPE_CYLINDER);
                sc_.nodes[node].params[0] = r;
                sc_.nodes[node].params[1] = h;
                if (finish_shape(node)) return node;
            }
        }
        reset();
        if (lit("torus")) {
            ws();
            if (lit("{")) {
                ws();
                double r1, r2;
                if (keyed_float("radius", r1)) {
                    ws();
                    if (keyed_float("radius", r2)) {
                        ws();
                        int node = new_node(RT_SHAPE_TORUS);
                        sc_.nodes[node].params[0] = r1;
                        sc_.nodes[node].params[1] = r2;
                        if (finish_shape(node)) return node;
                    }
                }
            }
        }
        reset();
        if (lit("cube")) {
            ws();
            if (lit("{")) {
                ws();
                double p1[3], p2[3];
                if (vec3(p1)) {
                    ws();
                    if (vec3(p2)) {
                        ws();
                        int node = new_node(RT_SHAPE_CUBE);
                        double *p = sc_.nodes[node].params;
                        for (int k = 0; k < 3; ++k) {
                            p[k] = p1[k];
                            p[3 + k] = p2[k];
                        }
                        if (finish_shape(node)) return node;
                    }
                }
            }
        }
        reset();
        const char *csg_names[3] = {"union", "intersection", "difference"};
        const int csg_kinds[3] = {RT_SHAPE_UNION, RT_SHAPE_INTERSECTION, RT_SHAPE_DIFFERENCE};
        for (int k = 0; k < 3; ++k) {
            if (lit(csg_names[k])) {
                ws();
                if (lit("{")) {
                    ws();
                    int a = solid_shape();
                    if (a >= 0) {
                        ws();
                        int b = solid_shape();
                        if (b >= 0) {
                            ws();
                            int node = new_node(csg_kinds[k]);
                            sc_.nodes[node].child_a = a;
                            sc_.nodes[node].child_b = b;
                            if (finish_shape(node)) return node;
                        }
                    }
                }
            }
            reset();
        }
        if (lit("homogenous_medium")) {
            ws();
            if (lit("{")) {
                ws();
                double density;
                if (keyed_float("density", density)) {
                    ws(); // permissive: absent in the reference rule
                    int b = solid_shape();
                    if (b >= 0) {
                        ws(); // permissive
                        int node = new_node(RT_SHAPE_MEDIUM);
                        sc_.nodes[node].child_a = b;
                        sc_.nodes[node].params[0] = density;
                        if (finish_shape(node)) return node;
                    }
                }
            }
        }
        reset();
        return -1;
    }

    // sdl_grammar.rs:126-131, sdl.rs:55-92
    int mesh()
    {
        size_t save = pos_;
        size_t nn = sc_.nodes.size(), nc = sc_.child_list.size(), nm = sc_.meshes.size();
        if (lit("mesh")) {
            ws();
            if (lit("{")) {
                ws();
                std::string path;
                if (lit("file")) {
                    ws();
                    if (string_lit(path)) {
                        ws(); // permissive (string() already ate plain whitespace)
                        // the file is read only once the rule is known to be a mesh
                        std::vector<int32_t> mesh_nodes;
                        load_obj(sc_, join_path(base_, path), mesh_nodes);
                        int node = new_node(RT_SHAPE_COMPOSITE);
                        sc_.nodes[node].child_a = (int32_t)sc_.child_list.size();
                        sc_.nodes[node].child_b = (int32_t)mesh_nodes.size();
                        for (int32_t m : mesh_nodes) sc_.child_list.push_back(m);
                        if (finish_shape(node)) return node;
                    }
                }
            }
        }
        pos_ = save;
        drop_nodes(nn, nc, nm);
        return -1;
    }

    // sdl_grammar.rs:236-257
    int texture()
    {
        size_t save = pos_;
        if (lit("texture")) {
            ws();
            if (lit("{")) {
                ws();
                rt_texture t;
                std::memset(&t, 0, sizeof(t));
                t.image = -1;
                bool ok = false;
                size_t s2 = pos_;
                if (lit("solid")) {
                    ws();
                    if (color(t.c1)) {
                        t.kind = RT_TEX_SOLID;
                        ok = true;
                    }
                }
                if (!ok) {
                    pos_ = s2;
                    if (lit("pattern")) {
                        ws();
                        if (lit("{")) {
                            ws();
                            if (lit("checkerboard")) {
                                ws();
                                if (color(t.c1)) {
                                    ws();
                                    if (color(t.c2)) {
                                        ws();
                                        if (number(t.scale)) {
                                            ws();
                                            if (lit("}")) {
                                                t.kind = RT_TEX_CHECKERBOARD;
                                                ok = true;
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
                if (!ok) {
                    pos_ = s2;
                    if (lit("image")) {
                        ws();
                        std::string path;
                        if (string_lit(path)) {
                            ws();
                            if (number(t.scale)) {
                                t.kind = RT_TEX_IMAGE;
                                t.image = load_image(sc_, join_path(base_, path));
                                ok = true;
                            }
                        }
                    }
                }
                if (ok) {
                    ws();
                    if (lit("}")) {
                        sc_.textures.push_back(t);
                        return (int)sc_.textures.size() - 1;
                    }
                }
            }
        }
        pos_ = save;
        return -1;
    }

    // sdl_grammar.rs:56,158-194
    int object_material()
    {
        size_t save = pos_;
        if (lit("material")) {
            ws();
            if (lit("{")) {
                ws();
                rt_material m;
                std::memset(&m, 0, sizeof(m));
                m.texture = -1;
                bool ok = false;
                size_t s2 = pos_;
                if (lit("lambertian")) {
                    ws();
                    m.texture = texture();
                    if (m.texture >= 0) {
                        m.kind = RT_MAT_LAMBERTIAN;
                        ok = true;
                    }
                }
                if (!ok) {
                    pos_ = s2;
                    if (lit("metal")) {
                        ws();
                        if (keyed_float("fuzz", m.fuzz)) {
                            ws();
                            m.texture = texture();
                            if (m.texture >= 0) {
                                m.kind = RT_MAT_METAL;
                                ok = true;
                            }
                        }
                    }
                }
                if (!ok) {
                    pos_ = s2;
                    m.fuzz = 0.0;
                    if (lit("dielectric")) {
                        ws();
                        if (keyed_float("ior", m.ior)) {
                            // `_ fuzz()?`: the whitespace is consumed even when fuzz is absent
                            ws();
                            double fz;
                            if (keyed_float("fuzz", fz)) m.fuzz = fz;
                            m.kind = RT_MAT_DIELECTRIC;
                            m.texture = -1;
                            ok = true;
                        }
                    }
                }
                if (!ok) {
                    pos_ = s2;
                    if (lit("diffuse_light")) {
                        ws();
                        if (keyed_float("intensity", m.intensity)) {
                            ws();
                            m.texture = texture();
                            if (m.texture >= 0) {
                                m.kind = RT_MAT_DIFFUSE_LIGHT;
                                ok = true;
                            }
                        }
                    }
                }
                if (!ok) {
                    pos_ = s2;
                    if (lit("isotropic")) {
                        ws();
                        m.texture = texture();
                        if (m.texture >= 0) {
                            m.kind = RT_MAT_ISOTROPIC;
                            ok = true;
                        }
                    }
                }
                if (ok) {
                    ws();
                    if (lit("}")) {
                        sc_.materials.push_back(m);
                        return (int)sc_.materials.size() - 1;
                    }
                }
            }
        }
        pos_ = save;
        return -1;
    }
};

} // namespace rt_host

using namespace rt_host;

extern "C" int rt_sdl_parse(const char *text, const char *base_dir, uint32_t width, uint32_t height,
                            rt_scene_desc **out)
{
    if (!text || !out) {
        set_last_error("rt_sdl_parse: null argument");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    SceneOwner *sc = new SceneOwner();
    try {
        Parser p(text, base_dir ? base_dir : "", width, height, *sc);
        p.scene();
        sc->finalize();
    } catch (const ParseError &e) {
        delete sc;
        set_last_error(e.msg);
        return e.status;
    } catch (const std::exception &e) {
        delete sc;
        set_last_error(std::string("rt_sdl_parse: ") + e.what());
        return RT_ERR_INVALID;
    }
    *out = &sc->desc;
    return RT_OK;
}

extern "C" int rt_sdl_parse_file(const char *path, const char *base_dir, uint32_t width, uint32_t height,
                                 rt_scene_desc **out)
{
    if (!path || !out) {
        set_last_error("rt_sdl_parse_file: null argument");
        return RT_ERR_INVALID;
    }
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        set_last_error(std::string("could not open scene file: ") + path);
        return RT_ERR_IO;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    std::string text = ss.str();
    return rt_sdl_parse(text.c_str(), base_dir, width, height, out);
}

extern "C" void rt_scene_desc_free(rt_scene_desc *desc)
{
    if (desc) delete reinterpret_cast<SceneOwner *>(desc);
}
