// lf_collada.cpp -- see lf_collada.h.  Own XML reader + COLLADA subset + half-edge connectivity;
// every step cites the reference code whose *result* it has to reproduce.
#include "lf_collada.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <utility>

namespace lfamd {
namespace {

struct Fail : std::runtime_error { using std::runtime_error::runtime_error; };

// ================================================================================ XML ==========
// What the reference gets from tinyxml2 (CGL/src/tinyxml2.cpp): element tree, attributes, and
// GetText() = the value of an element's FIRST child if that child is a text node.  White space in
// front of a node is skipped, so an element whose first child is another element has no text.
struct Xml {
  std::string name;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<std::unique_ptr<Xml>> kids;   // child elements, document order
  bool has_text = false;                    // first child is text
  std::string text;
  Xml* parent = nullptr;
  size_t pos_in_parent = 0;

  const char* attr(const char* k) const {
    for (auto& a : attrs) if (a.first == k) return a.second.c_str();
    return nullptr;
  }
  Xml* first(const char* tag = nullptr) const {
    for (auto& c : kids) if (!tag || c->name == tag) return c.get();
    return nullptr;
  }
  Xml* next(const char* tag = nullptr) const {   // NextSiblingElement
    if (!parent) return nullptr;
    for (size_t i = pos_in_parent + 1; i < parent->kids.size(); i++)
      if (!tag || parent->kids[i]->name == tag) return parent->kids[i].get();
    return nullptr;
  }
  const std::string& get_text(const char* what) const {
    if (!has_text) throw Fail(std::string("element <") + name + "> has no text (" + what + ")");
    return text;
  }
  int int_attr(const char* k) const {   // IntAttribute: sscanf("%d"), 0 when absent
    const char* v = attr(k);
    int r = 0;
    if (v) std::sscanf(v, "%d", &r);
    return r;
  }
};

class XmlReader {
 public:
  explicit XmlReader(const std::string& s) : s_(s) {}
  std::unique_ptr<Xml> parse_document() {
    std::unique_ptr<Xml> root;
    for (;;) {
      skip_ws();
      if (p_ >= s_.size()) break;
      if (starts("<?")) { skip_until("?>"); continue; }
      if (starts("<!--")) { skip_until("-->"); continue; }
      if (starts("<!")) { skip_until(">"); continue; }
      if (s_[p_] == '<') {
        auto e = parse_element(nullptr, 0);
        if (!root) root = std::move(e);
        continue;
      }
      throw Fail("XML: text outside the root element");
    }
    if (!root) throw Fail("XML: no root element");
    return root;
  }

 private:
  const std::string& s_;
  size_t p_ = 0;
  bool starts(const char* t) const { return s_.compare(p_, std::strlen(t), t) == 0; }
  void skip_ws() { while (p_ < s_.size() && std::isspace((unsigned char)s_[p_])) p_++; }
  void skip_until(const char* t) {
    size_t q = s_.find(t, p_);
    if (q == std::string::npos) throw Fail("XML: unterminated construct");
    p_ = q + std::strlen(t);
  }
  static bool name_char(char c) { return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.'; }
  std::string parse_name() {
    size_t b = p_;
    while (p_ < s_.size() && name_char(s_[p_])) p_++;
    if (p_ == b) throw Fail("XML: name expected");
    return s_.substr(b, p_ - b);
  }
  static std::string decode(const std::string& in) {   // the five predefined entities + &#n;
    std::string out;
    for (size_t i = 0; i < in.size(); i++) {
      if (in[i] != '&') { out += in[i]; continue; }
      size_t e = in.find(';', i);
      if (e == std::string::npos) { out += in[i]; continue; }
      std::string ent = in.substr(i + 1, e - i - 1);
      if (ent == "lt") out += '<';
      else if (ent == "gt") out += '>';
      else if (ent == "amp") out += '&';
      else if (ent == "quot") out += '"';
      else if (ent == "apos") out += '\'';
      else if (!ent.empty() && ent[0] == '#') {
        long v = ent.size() > 1 && (ent[1] == 'x' || ent[1] == 'X') ? std::strtol(ent.c_str() + 2, nullptr, 16)
                                                                     : std::strtol(ent.c_str() + 1, nullptr, 10);
        if (v > 0 && v < 128) out += (char)v;
      } else { out += in.substr(i, e - i + 1); }
      i = e;
    }
    return out;
  }
  std::unique_ptr<Xml> parse_element(Xml* parent, size_t pos) {
    std::unique_ptr<Xml> e(new Xml());
    e->parent = parent;
    e->pos_in_parent = pos;
    p_++;  // '<'
    e->name = parse_name();
    for (;;) {
      skip_ws();
      if (p_ >= s_.size()) throw Fail("XML: unterminated tag");
      if (starts("/>")) { p_ += 2; return e; }
      if (s_[p_] == '>') { p_++; break; }
      std::string k = parse_name();
      skip_ws();
      if (p_ >= s_.size() || s_[p_] != '=') throw Fail("XML: '=' expected in <" + e->name + ">");
      p_++;
      skip_ws();
      char q = p_ < s_.size() ? s_[p_] : 0;
      if (q != '"' && q != '\'') throw Fail("XML: quoted attribute value expected");
      size_t b = ++p_;
      while (p_ < s_.size() && s_[p_] != q) p_++;
      if (p_ >= s_.size()) throw Fail("XML: unterminated attribute");
      e->attrs.emplace_back(k, decode(s_.substr(b, p_ - b)));
      p_++;
    }
    bool first_child = true;
    for (;;) {
      skip_ws();
      if (p_ >= s_.size()) throw Fail("XML: unterminated element <" + e->name + ">");
      if (starts("</")) {
        p_ += 2;
        std::string n = parse_name();
        if (n != e->name) throw Fail("XML: </" + n + "> closes <" + e->name + ">");
        skip_ws();
        if (p_ >= s_.size() || s_[p_] != '>') throw Fail("XML: '>' expected");
        p_++;
        return e;
      }
      if (starts("<!--")) { skip_until("-->"); first_child = false; continue; }
      if (starts("<![CDATA[")) {
        size_t b = p_ + 9;
        skip_until("]]>");
        if (first_child) { e->has_text = true; e->text = s_.substr(b, p_ - 3 - b); }
        first_child = false;
        continue;
      }
      if (starts("<?")) { skip_until("?>"); first_child = false; continue; }
      if (s_[p_] == '<') {
        e->kids.push_back(parse_element(e.get(), e->kids.size()));
        first_child = false;
        continue;
      }
      size_t b = p_;
      while (p_ < s_.size() && s_[p_] != '<') p_++;
      if (first_child) { e->has_text = true; e->text = decode(s_.substr(b, p_ - b)); }
      first_child = false;
    }
  }
};

// ================================================================================ maths ========
typedef ColladaVec3 V3;
struct V4 { double x, y, z, w; };
struct M4 { double m[4][4]; };   // m[i][j] = entry (row i, column j)

M4 identity() { M4 r{}; for (int i = 0; i < 4; i++) r.m[i][i] = 1.0; return r; }
// Matrix4x4::operator*(Matrix4x4) as the reference's AVX build computes it: entry (i, j) is the dot
// product of COLUMN i of A with column j of B (CGL/src/matrix4x4.cpp:131-133, vector4D.h:257-259)
M4 mul_ref(const M4& A, const M4& B) {
  M4 C;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      C.m[i][j] = A.m[0][i] * B.m[0][j] + A.m[1][i] * B.m[1][j] + A.m[2][i] * B.m[2][j] + A.m[3][i] * B.m[3][j];
  return C;
}
// Matrix4x4::operator*(Vector4D): x0*col0 + x1*col1 + x2*col2 + x3*col3 (matrix4x4.cpp:146-149)
V4 mul(const M4& A, const V4& v) {
  V4 r;
  r.x = ((v.x * A.m[0][0] + v.y * A.m[0][1]) + v.z * A.m[0][2]) + v.w * A.m[0][3];
  r.y = ((v.x * A.m[1][0] + v.y * A.m[1][1]) + v.z * A.m[1][2]) + v.w * A.m[1][3];
  r.z = ((v.x * A.m[2][0] + v.y * A.m[2][1]) + v.z * A.m[2][2]) + v.w * A.m[2][3];
  r.w = ((v.x * A.m[3][0] + v.y * A.m[3][1]) + v.z * A.m[3][2]) + v.w * A.m[3][3];
  return r;
}
V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
V3 to3D(const V4& v) { return v3(v.x, v.y, v.z); }
V3 project3D(const V4& v) { double iw = 1.0 / v.w; return v3(v.x * iw, v.y * iw, v.z * iw); }  // vector4D.cpp:14-17
V3 sub(const V3& a, const V3& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
V3 neg(const V3& a) { return v3(-a.x, -a.y, -a.z); }
V3 scale(const V3& a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
double norm(const V3& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }          // vector3D.h:193-199
V3 unit(const V3& a) { double r = 1.0 / norm(a); return scale(a, r); }                       // :215-218
V3 cross(const V3& u, const V3& v) {                                                          // :265-269
  return v3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}

V3 spectrum_from(const std::string& s) {   // collada.cpp:30-41: three doubles, missing ones stay 0
  V3 r;
  std::stringstream ss(s);
  ss >> r.x; ss >> r.y; ss >> r.z;
  return r;
}

// ================================================================================ parser =======
struct Loader {
  ColladaScene& out;
  std::map<std::string, Xml*> ids;   // uri table (collada.cpp:55-68): last id wins
  V3 up = v3(0, 1, 0);
  M4 transform = identity();

  explicit Loader(ColladaScene& o) : out(o) {}

  void uri_load(Xml* e) {
    if (const char* id = e->attr("id")) ids[id] = e;
    for (auto& c : e->kids) uri_load(c.get());
  }
  Xml* uri_find(const std::string& id) const {
    auto it = ids.find(id);
    return it == ids.end() ? nullptr : it->second;
  }
  // get_element (collada.cpp:77-97): first-child walk along a/b/c, then ONE url indirection
  Xml* get_element(Xml* xml, const std::string& query) const {
    Xml* e = xml;
    std::stringstream ss(query);
    std::string tok;
    while (e && std::getline(ss, tok, '/')) e = e->first(tok.c_str());
    if (e) if (const char* url = e->attr("url")) e = uri_find(std::string(url + 1));
    return e;
  }
  Xml* technique_common(Xml* xml) const {   // :100-114
    if (Xml* prof = xml->first("profile_COMMON"))
      for (Xml* t = prof->first("technique"); t; t = t->next("technique")) {
        const char* sid = t->attr("sid");
        if (!sid) throw Fail("technique without sid (the reference dereferences it)");
        if (std::string(sid) == "common") return t;
      }
    return xml->first("technique_common");
  }
  Xml* technique_cgl(Xml* xml) const {      // :117-129
    for (Xml* t = get_element(xml, "extra/technique"); t; t = t->next("technique")) {
      const char* prof = t->attr("profile");
      if (!prof) throw Fail("technique without profile (the reference dereferences it)");
      if (std::string(prof) == "CGL") return t;
    }
    return nullptr;
  }

  void load(Xml* root) {
    if (root->name != "COLLADA") throw Fail("not a COLLADA file");
    uri_load(root);
    if (Xml* asset = get_element(root, "asset")) {   // :160-199
      Xml* up_axis = get_element(asset, "up_axis");
      if (!up_axis) throw Fail("no up direction defined in COLLADA file");
      const std::string& dir = up_axis->get_text("up_axis");
      transform = identity();
      if (dir == "X_UP") {
        transform.m[0][0] = 0; transform.m[0][1] = 1; transform.m[1][0] = 1; transform.m[1][1] = 0;
        transform.m[2][2] = -1;
        up = v3(1, 0, 0);
      } else if (dir == "Z_UP") {
        transform.m[1][1] = 0; transform.m[1][2] = 1; transform.m[2][1] = 1; transform.m[2][2] = 0;
        transform.m[0][0] = -1;
        up = v3(0, 0, 1);
      } else if (dir == "Y_UP") {
        up = v3(0, 1, 0);
      } else {
        throw Fail("invalid up direction in COLLADA file");
      }
    }
    Xml* scene = get_element(root, "scene/instance_visual_scene");   // :205-217
    if (!scene) throw Fail("no scene description found");
    for (Xml* n = get_element(scene, "node"); n; n = n->next("node")) parse_node(n);
  }

  void parse_node(Xml* xml) {   // :226-429
    if (!xml->attr("id") || !xml->attr("name")) throw Fail("node without id/name (the reference dereferences them)");
    M4 node_t = identity();
    for (Xml* e = xml->first(); e; e = e->next()) {
      if (e->name == "matrix") {
        std::stringstream ss(e->get_text("matrix"));
        M4 m{};
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) ss >> m.m[i][j];
        node_t = m;
        break;   // the reference stops reading transform elements here
      }
      if (e->name == "rotate" || e->name == "translate" || e->name == "scale")
        throw Fail("<" + e->name + "> node transforms have no defined result in the reference "
                   "(uninitialised Matrix4x4, collada.cpp:262-318); use <matrix>");
    }
    const M4 saved = transform;
    node_t = mul_ref(transform, node_t);
    transform = node_t;
    for (Xml* c = get_element(xml, "node"); c; c = c->next("node")) parse_node(c);
    transform = saved;

    Xml* e_camera = get_element(xml, "instance_camera");
    Xml* e_light = get_element(xml, "instance_light");
    Xml* e_geometry = get_element(xml, "instance_geometry");
    if (e_camera) {
      add_camera(e_camera, node_t);
    } else if (e_light) {
      add_light(e_light, node_t);
    } else if (e_geometry) {
      int material = -1;
      auto find_material = [&]() {
        Xml* im = get_element(xml, "instance_geometry/bind_material/technique_common/instance_material");
        if (!im) return;
        const char* target = im->attr("target");
        if (!target) throw Fail("no target material in instance_material");
        Xml* mat = uri_find(std::string(target + 1));
        if (!mat) throw Fail(std::string("invalid target material id: ") + (target + 1));
        material = parse_material(mat);
      };
      if (get_element(e_geometry, "mesh")) {
        find_material();
        add_polymesh(e_geometry, node_t, material);
      } else if (get_element(e_geometry, "extra")) {
        find_material();
        add_sphere(e_geometry, node_t, material);
      } else {
        throw Fail("geometry without mesh or extra: the reference leaves the node without an instance");
      }
    } else {
      throw Fail(std::string("node '") + xml->attr("id") + "' has no camera/light/geometry instance "
                 "(Application::load dereferences a null instance)");
    }
  }

  static double atof_text(Xml* e, const char* what) { return std::atof(e->get_text(what).c_str()); }

  void add_camera(Xml* xml, const M4& T) {   // parse_camera :431-475 + Application::load :249-255
    if (!xml->attr("id") || !xml->attr("name")) throw Fail("camera without id/name");
    Xml* persp = get_element(xml, "optics/technique_common/perspective");
    if (!persp) throw Fail("no perspective defined in camera");
    Xml *xf = persp->first("xfov"), *yf = persp->first("yfov"), *zn = persp->first("znear"), *zf = persp->first("zfar");
    float hFov = xf ? (float)atof_text(xf, "xfov") : 50.0f;
    float vFov = yf ? (float)atof_text(yf, "yfov") : 35.0f;
    float nClip = zn ? (float)atof_text(zn, "znear") : 0.001f;
    float fClip = zf ? (float)atof_text(zf, "zfar") : 1000.0f;
    if (!yf) {
      Xml* ar = get_element(persp, "aspect_ratio");
      if (!ar) throw Fail("incomplete perspective definition");
      float aspect = (float)atof_text(ar, "aspect_ratio");
      const double pi = 3.14159265358979323846;
      // 2 * degrees(atan(tan(radians(0.5 * hFov)) / aspect_ratio)), CGL/misc.h radians/degrees
      double half = (0.5 * hFov) * (pi / 180.0);
      vFov = (float)(2 * ((std::atan(std::tan(half) / aspect)) * (180.0 / pi)));
    }
    ColladaCamera& c = out.camera;
    // Application::load keeps ONE running c_pos across camera nodes (starts at the origin)
    V3 p = c.present ? c.pos : v3(0, 0, 0);
    c.pos = to3D(mul(T, V4{p.x, p.y, p.z, 1.0}));
    c.dir = unit(to3D(mul(T, V4{0.0, 0.0, -1.0, 1.0})));
    c.up = up;
    c.hFov = hFov; c.vFov = vFov; c.nClip = nClip; c.fClip = fClip;
    c.present = true;
    out.items.push_back({ColladaItem::CAMERA, (int)out.camera_nodes.size(), 1});
    out.camera_nodes.push_back(c);
  }

  void add_light(Xml* xml, const M4& T) {   // parse_light :477-580 + init_light + get_static_light
    if (!xml->attr("id") || !xml->attr("name")) throw Fail("light without id/name");
    Xml* tc = technique_common(xml);
    Xml* tg = technique_cgl(xml);
    Xml* tech = tg ? tg : tc;
    if (!tech) throw Fail("no supported profile defined in light");
    Xml* e = tech->first();
    if (!e) throw Fail("light technique without a light element (LightType::NONE: Application::load stores a null light)");
    const std::string& type = e->name;
    V3 spectrum = v3(1, 1, 1);
    auto color = [&](const char* what) {
      Xml* c = get_element(e, "color");
      if (!c) throw Fail(std::string("no color definition in ") + what + " light");
      spectrum = spectrum_from(c->get_text("color"));
    };
    const V3 info_pos = v3(0, 0, 0), info_dir = v3(0, 0, -1), info_up = v3(0, 1, 0);   // light_info.cpp:13-15
    ColladaLight L;
    if (type == "ambient") {
      color("ambient");
      L.type = LIGHT_HEMISPHERE;           // AmbientLight -> InfiniteHemisphereLight(spectrum)
    } else if (type == "directional") {
      color("directional");
      L.type = LIGHT_DIRECTIONAL;
      // GLScene::DirectionalLight (directional_light.h:14-31): position = -(T (direction, 1)),
      // direction = position.unit(); SceneObjects::DirectionalLight (light.cpp:11-16):
      // posLight = -position, dirToLight = -direction.unit()
      V3 position = neg(to3D(mul(T, V4{info_dir.x, info_dir.y, info_dir.z, 1.0})));
      V3 direction = unit(position);
      L.position = neg(position);
      L.direction = neg(unit(direction));
    } else if (type == "area") {
      color("area");
      L.type = LIGHT_AREA;                 // area_light.h:13-27
      V3 position = to3D(mul(T, V4{info_pos.x, info_pos.y, info_pos.z, 1.0}));
      V3 d = sub(to3D(mul(T, V4{info_dir.x, info_dir.y, info_dir.z, 1.0})), position);
      double rn = 1. / norm(d);            // normalize(): *= (1/norm)
      L.position = position;
      L.direction = scale(d, rn);
      V3 dim_y = info_up, dim_x = cross(info_up, info_dir);
      L.dim_x = sub(to3D(mul(T, V4{dim_x.x, dim_x.y, dim_x.z, 1.0})), position);
      L.dim_y = sub(to3D(mul(T, V4{dim_y.x, dim_y.y, dim_y.z, 1.0})), position);
    } else if (type == "point") {
      color("point");
      if (!get_element(e, "constant_attenuation") || !get_element(e, "linear_attenuation") ||
          !get_element(e, "quadratic_attenuation"))
        throw Fail("incomplete definition of point light");
      L.type = LIGHT_POINT;
      L.position = to3D(mul(T, V4{info_pos.x, info_pos.y, info_pos.z, 1.0}));
    } else if (type == "spot") {
      color("spot");
      if (!e->first("falloff_angle") || !e->first("falloff_exponent") || !get_element(e, "constant_attenuation") ||
          !get_element(e, "linear_attenuation") || !get_element(e, "quadratic_attenuation"))
        throw Fail("incomplete definition of spot light");
      L.type = LIGHT_SPOT;
      L.position = to3D(mul(T, V4{info_pos.x, info_pos.y, info_pos.z, 1.0}));
      V3 d = sub(to3D(mul(T, V4{info_dir.x, info_dir.y, info_dir.z, 1.0})), L.position);
      L.direction = scale(d, 1. / norm(d));
    } else {
      throw Fail("light type " + type + " is not supported");
    }
    L.radiance = spectrum;
    out.items.push_back({ColladaItem::LIGHT, (int)out.lights.size(), 1});
    out.lights.push_back(L);
  }

  int add_material(int kind, const V3& rgb) {
    ColladaMaterial m; m.kind = kind; m.rgb = rgb;
    out.materials.push_back(m);
    return (int)out.materials.size() - 1;
  }
  int default_material() { return add_material(BSDF_DIFFUSE, v3(0.5f, 0.5f, 0.5f)); }

  int parse_material(Xml* xml) {   // :862-950
    if (!xml->attr("id") || !xml->attr("name")) throw Fail("material without id/name");
    Xml* effect = get_element(xml, "instance_effect");
    if (!effect) throw Fail("no target effects found for material");
    Xml* tc = technique_common(effect);
    Xml* tg = technique_cgl(effect);
    if (tg) {
      int result = -1;
      for (Xml* b = tg->first(); b; b = b->next()) {
        auto need = [&](const char* child) {
          Xml* c = get_element(b, child);
          if (!c) throw Fail("<" + b->name + "> without <" + child + ">");
          return c;
        };
        if (b->name == "emission") result = add_material(BSDF_EMISSION, spectrum_from(need("radiance")->get_text("radiance")));
        else if (b->name == "mirror") { need("reflectance"); result = add_material(BSDF_MIRROR, v3(0, 0, 0)); }
        else if (b->name == "microfacet") { need("alpha"); need("eta"); need("k"); result = add_material(BSDF_MICROFACET, v3(0, 0, 0)); }
        else if (b->name == "refraction") { need("transmittance"); need("roughness"); need("ior"); result = add_material(BSDF_REFRACTION, v3(0, 0, 0)); }
        else if (b->name == "glass") { need("transmittance"); need("reflectance"); need("roughness"); need("ior"); result = add_material(BSDF_GLASS, v3(0, 0, 0)); }
      }
      if (result < 0) throw Fail("CGL material technique without a known BSDF (the reference leaves bsdf uninitialised)");
      return result;
    }
    if (tc) {
      Xml* phong = get_element(tc, "phong/diffuse/color");
      Xml* lambert = get_element(tc, "lambert/diffuse/color");
      if (lambert) return add_material(BSDF_DIFFUSE, spectrum_from(lambert->get_text("color")));
      if (phong) return add_material(BSDF_DIFFUSE, spectrum_from(phong->get_text("color")));
      return default_material();
    }
    return default_material();
  }

  void add_sphere(Xml* xml, const M4& T, int material) {   // parse_sphere :582-604, init_sphere, GLScene::Sphere
    if (!xml->attr("id") || !xml->attr("name")) throw Fail("geometry without id/name");
    Xml* tech = technique_cgl(xml);
    if (!tech) throw Fail("no CGL profile technique in sphere geometry");
    Xml* radius = get_element(tech, "sphere/radius");
    if (!radius) throw Fail("invalid sphere definition");
    float r = (float)std::atof(radius->get_text("radius").c_str());
    ColladaSphere s;
    s.o = project3D(mul(T, V4{0, 0, 0, 1}));
    double sc = norm(to3D(mul(T, V4{1, 0, 0, 0})));
    s.r = r * sc;
    s.material = material >= 0 ? material : default_material();
    out.items.push_back({ColladaItem::SPHERE, (int)out.spheres.size(), 1});
    out.spheres.push_back(s);
  }

  void add_polymesh(Xml* xml, const M4& T, int material) {   // parse_polymesh :607-859
    if (!xml->attr("id") || !xml->attr("name")) throw Fail("geometry without id/name");
    Xml* mesh = xml->first("mesh");
    if (!mesh) throw Fail("no mesh data defined in geometry");
    std::map<std::string, std::vector<float>> sources;
    for (Xml* s = mesh->first("source"); s; s = s->next("source")) {
      const char* id = s->attr("id");
      if (!id) throw Fail("source without id");
      if (Xml* fa = s->first("float_array")) {
        std::vector<float> fl;
        const std::string& txt = fa->get_text("float_array");
        const char* c = txt.c_str();
        size_t n = (size_t)fa->int_attr("count");
        float last = 0.0f;   // `ss >> f` leaves f unchanged once the stream fails
        for (size_t i = 0; i < n; i++) {
          char* end = nullptr;
          float f = std::strtof(c, &end);
          if (end == c) { fl.push_back(last); continue; }
          last = f; c = end;
          fl.push_back(f);
        }
        sources[id] = fl;
      }
    }
    Xml* verts = mesh->first("vertices");
    if (!verts) throw Fail("no vertices defined in geometry");
    if (!verts->attr("id")) throw Fail("vertices without id");
    const std::string vertices_id = verts->attr("id");
    std::vector<V3> vertices;
    for (Xml* in = verts->first("input"); in; in = in->next("input")) {
      const char* sem = in->attr("semantic");
      if (!sem) throw Fail("input without semantic");
      if (std::string(sem) == "POSITION") {
        const char* src = in->attr("source");
        if (!src) throw Fail("input without source");
        auto it = sources.find(src + 1);
        if (it == sources.end()) throw Fail(std::string("undefined input source: ") + (src + 1));
        const std::vector<float>& fl = it->second;
        if (fl.size() % 3) throw Fail("POSITION array length is not a multiple of 3 (the reference reads past the end)");
        for (size_t i = 0; i < fl.size(); i += 3) vertices.push_back(v3(fl[i], fl[i + 1], fl[i + 2]));
      }
    }
    Xml* plist = mesh->first("polylist");
    bool is_triangles = false;
    if (!plist) { plist = mesh->first("triangles"); is_triangles = true; }
    std::vector<std::vector<size_t>> polygons;
    std::vector<V3> mesh_vertices;
    if (plist) {
      bool has_v = false, has_n = false, has_t = false;
      size_t off_v = 0;
      for (Xml* in = plist->first("input"); in; in = in->next("input")) {
        const char* sem = in->attr("semantic");
        const char* src = in->attr("source");
        if (!sem || !src) throw Fail("polylist input without semantic/source");
        size_t offset = (size_t)in->int_attr("offset");
        std::string semantic = sem, source = src + 1;
        if (semantic == "VERTEX") {
          has_v = true; off_v = offset;
          if (source != vertices_id) throw Fail("undefined source for VERTEX semantic: " + source);
          mesh_vertices = vertices;
        }
        if (semantic == "NORMAL") {
          has_n = true;
          if (!sources.count(source)) throw Fail("undefined source for NORMAL semantic: " + source);
        }
        if (semantic == "TEXCOORD") {
          has_t = true;
          if (!sources.count(source)) throw Fail("undefined source for TEXCOORD semantic: " + source);
        }
      }
      size_t n_poly = (size_t)plist->int_attr("count");
      size_t stride = (has_v ? 1 : 0) + (has_n ? 1 : 0) + (has_t ? 1 : 0);   // :766-768 (not max offset + 1)
      std::vector<size_t> sizes;
      size_t n_idx = 0;
      if (!is_triangles) {
        Xml* vc = plist->first("vcount");
        if (!vc) throw Fail("polygon sizes undefined in geometry");
        std::stringstream ss(vc->get_text("vcount"));
        size_t sz = 0;
        for (size_t i = 0; i < n_poly; i++) { ss >> sz; sizes.push_back(sz); n_idx += sz * stride; }
      } else {
        for (size_t i = 0; i < n_poly; i++) { sizes.push_back(3); n_idx += 3 * stride; }
      }
      Xml* pe = plist->first("p");
      if (!pe) throw Fail("no index array defined in geometry");
      std::vector<size_t> idx;
      {
        std::stringstream ss(pe->get_text("p"));
        size_t v = 0;
        for (size_t i = 0; i < n_idx; i++) { ss >> v; idx.push_back(v); }
      }
      polygons.resize(n_poly);
      if (has_v) {
        size_t k = 0;
        for (size_t i = 0; i < n_poly; i++)
          for (size_t j = 0; j < sizes[i]; j++, k++) {
            if (k * stride + off_v >= idx.size()) throw Fail("index array too short for its inputs (the reference reads past the end)");
            polygons[i].push_back(idx[k * stride + off_v]);
          }
      }
    }
    // GLScene::Mesh::Mesh (mesh.cpp:21-45): positions through the node transform, then the half-edge mesh
    for (V3& v : mesh_vertices) v = project3D(mul(T, V4{v.x, v.y, v.z, 1.0}));
    const int mat = material >= 0 ? material : default_material();
    const int first = (int)out.triangles.size();
    build_triangles(polygons, mesh_vertices, mat);
    out.items.push_back({ColladaItem::MESH, first, (int)out.triangles.size() - first});
  }

  // HalfedgeMesh::build (halfEdgeMesh.cpp:28-404) + Vertex::computeNormal (halfEdgeMesh.h:492-515) +
  // SceneObjects::Mesh::Mesh (object.cpp:14-45), on index arrays instead of linked lists
  void build_triangles(const std::vector<std::vector<size_t>>& polygons, const std::vector<V3>& positions, int mat) {
    const int NONE = -1;
    struct HE { int next = -1, twin = -1, vertex = -1; bool boundary_face = false; };
    std::vector<HE> he;
    std::map<size_t, int> index_to_vertex;           // sorted by index
    std::vector<int> vert_he;                        // per vertex (first-appearance order)
    std::vector<size_t> vert_degree;
    for (auto& p : polygons) {
      if (p.size() < 3) throw Fail("each polygon must have at least three vertices");
      std::set<size_t> distinct;
      for (size_t i : p) {
        distinct.insert(i);
        auto it = index_to_vertex.find(i);
        if (it == index_to_vertex.end()) {
          index_to_vertex[i] = (int)vert_he.size();
          vert_he.push_back(NONE);
          vert_degree.push_back(1);
        } else {
          vert_degree[it->second]++;
        }
      }
      if (distinct.size() < p.size()) throw Fail("a polygon does not have distinct vertices");
    }
    std::vector<int> face_he(polygons.size(), NONE);
    std::map<std::pair<size_t, size_t>, int> pair_to_he;
    for (size_t f = 0; f < polygons.size(); f++) {
      const auto& p = polygons[f];
      const size_t deg = p.size();
      std::vector<int> ring;
      for (size_t i = 0; i < deg; i++) {
        size_t a = p[i], b = p[(i + 1) % deg];
        if (pair_to_he.count({a, b})) throw Fail("non-manifold or inconsistently oriented mesh (duplicate oriented edge)");
        int h = (int)he.size();
        he.push_back(HE());
        pair_to_he[{a, b}] = h;
        face_he[f] = h;                              // the face keeps the LAST half-edge of its polygon
        he[h].vertex = index_to_vertex[a];
        vert_he[he[h].vertex] = h;
        ring.push_back(h);
        auto tw = pair_to_he.find({b, a});
        if (tw != pair_to_he.end()) { he[h].twin = tw->second; he[tw->second].twin = h; }
      }
      for (size_t i = 0; i < deg; i++) he[ring[i]].next = ring[(i + 1) % deg];
    }
    const size_t n_vert = vert_he.size();
    // boundary vertices point at their half-edge without a twin (:232-243)
    for (size_t v = 0; v < n_vert; v++) {
      int h = vert_he[v];
      do {
        if (he[h].twin == NONE) { vert_he[v] = h; break; }
        h = he[he[h].twin].next;
      } while (h != vert_he[v]);
    }
    // boundary loops (:244-281); the loop also visits the half-edges it appends (they have twins)
    for (size_t h0 = 0; h0 < he.size(); h0++) {
      if (he[h0].twin != NONE) continue;
      std::vector<int> loop;
      int i = (int)h0;
      do {
        int t = (int)he.size();
        he.push_back(HE());
        loop.push_back(t);
        he[i].twin = t;
        he[t].twin = i;
        he[t].boundary_face = true;
        he[t].vertex = he[he[i].next].vertex;
        i = he[i].next;
        while (i != (int)h0 && he[i].twin != NONE) { i = he[i].twin; i = he[i].next; }
      } while (i != (int)h0);
      const size_t deg = loop.size();
      for (size_t p = 0; p < deg; p++) he[loop[p]].next = loop[(p + deg - 1) % deg];
    }
    for (size_t v = 0; v < n_vert; v++) vert_he[v] = he[he[vert_he[v]].twin].next;   // :282-284
    for (size_t v = 0; v < n_vert; v++) {   // manifold check (:285-305)
      size_t count = 0;
      int h = vert_he[v];
      do {
        if (!he[h].boundary_face) count++;
        h = he[he[h].twin].next;
      } while (h != vert_he[v]);
      if (count != vert_degree[v]) throw Fail("at least one of the vertices is nonmanifold");
    }
    if (positions.size() != n_vert)
      throw Fail("number of vertex positions is different from the number of distinct vertices");
    std::vector<V3> pos(n_vert);
    {
      size_t i = 0;   // the i-th smallest index takes the i-th position (:317-332)
      for (auto& kv : index_to_vertex) pos[kv.second] = positions[i++];
    }
    std::vector<V3> nrm(n_vert);
    for (size_t v = 0; v < n_vert; v++) {
      bool boundary = false;
      int h = vert_he[v];
      do {
        if (he[h].boundary_face) { boundary = true; break; }
        h = he[he[h].twin].next;
      } while (h != vert_he[v]);
      V3 n = v3(0, 0, 0);
      const V3 pi = pos[v];
      h = vert_he[v];
      do {
        const V3 pj = pos[he[he[h].next].vertex];
        const V3 pk = pos[he[he[he[h].next].next].vertex];
        const V3 c = cross(sub(pj, pi), sub(pk, pi));
        n = v3(n.x + c.x, n.y + c.y, n.z + c.z);
        h = boundary ? he[he[h].next].twin : he[he[h].twin].next;
      } while (h != vert_he[v]);
      nrm[v] = scale(n, 1. / norm(n));   // normalize(): *= (1/norm)
    }
    for (size_t f = 0; f < polygons.size(); f++) {
      int h = face_he[f];
      int a = he[h].vertex, b = he[he[h].next].vertex, c = he[he[he[h].next].next].vertex;
      ColladaTriangle t;
      t.p[0] = pos[a]; t.p[1] = pos[b]; t.p[2] = pos[c];
      t.n[0] = nrm[a]; t.n[1] = nrm[b]; t.n[2] = nrm[c];
      t.material = mat;
      out.triangles.push_back(t);
    }
  }
};

void put3(std::string& s, const char* tag, const V3& v) {
  char buf[128];
  std::snprintf(buf, sizeof buf, " %s %a %a %a", tag, v.x, v.y, v.z);
  s += buf;
}
void put_bsdf(std::string& s, const ColladaMaterial& m) {
  static const char* names[] = {"diffuse", "emission", "mirror", "glass", "refraction", "microfacet"};
  s += " bsdf ";
  s += names[m.kind];
  if (m.kind == BSDF_DIFFUSE || m.kind == BSDF_EMISSION) put3(s, "rgb", m.rgb);
}

}  // namespace

bool load_collada(const std::string& path, ColladaScene& out, std::string& error) {
  out = ColladaScene();
  try {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw Fail("cannot open " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();
    XmlReader reader(text);
    std::unique_ptr<Xml> root = reader.parse_document();
    Loader L(out);
    L.load(root.get());
  } catch (const std::exception& e) {
    error = e.what();
    out = ColladaScene();
    return false;
  }
  return true;
}

std::string dump_collada(const ColladaScene& sc) {
  std::string s;
  char buf[256];
  for (const ColladaItem& it : sc.items) {
    switch (it.kind) {
      case ColladaItem::CAMERA: {
        const ColladaCamera& c = sc.camera_nodes[it.index];
        std::snprintf(buf, sizeof buf, "camera hfov %a vfov %a nclip %a fclip %a", c.hFov, c.vFov, c.nClip, c.fClip);
        s += buf;
        put3(s, "pos", c.pos); put3(s, "dir", c.dir); put3(s, "up", c.up);
        s += "\n";
        break;
      }
      case ColladaItem::LIGHT: {
        const ColladaLight& l = sc.lights[it.index];
        switch (l.type) {
          case LIGHT_DIRECTIONAL:
            s += "light directional"; put3(s, "rad", l.radiance); put3(s, "dir_to_light", l.direction);
            put3(s, "pos_light", l.position); break;
          case LIGHT_POINT: s += "light point"; put3(s, "rad", l.radiance); put3(s, "pos", l.position); break;
          case LIGHT_AREA:
            s += "light area"; put3(s, "rad", l.radiance); put3(s, "pos", l.position); put3(s, "dir", l.direction);
            put3(s, "dim_x", l.dim_x); put3(s, "dim_y", l.dim_y); break;
          case LIGHT_HEMISPHERE: s += "light hemisphere"; put3(s, "rad", l.radiance); break;
          case LIGHT_SPOT: s += "light spot"; put3(s, "rad", l.radiance); put3(s, "pos", l.position); break;
        }
        s += "\n";
        break;
      }
      case ColladaItem::SPHERE: {
        const ColladaSphere& sp = sc.spheres[it.index];
        s += "sphere"; put3(s, "o", sp.o);
        std::snprintf(buf, sizeof buf, " r %a", sp.r);
        s += buf;
        put_bsdf(s, sc.materials[sp.material]);
        s += "\n";
        break;
      }
      case ColladaItem::MESH: {
        std::snprintf(buf, sizeof buf, "mesh %d", it.count);
        s += buf;
        // a mesh without triangles still has its material: the item remembers none, so look it up
        // through its first triangle, or fall back to the last material added
        const ColladaMaterial& m = it.count > 0 ? sc.materials[sc.triangles[it.index].material]
                                                : sc.materials.back();
        put_bsdf(s, m);
        s += "\n";
        for (int k = 0; k < it.count; k++) {
          const ColladaTriangle& t = sc.triangles[it.index + k];
          s += "tri";
          put3(s, "p1", t.p[0]); put3(s, "p2", t.p[1]); put3(s, "p3", t.p[2]);
          put3(s, "n1", t.n[0]); put3(s, "n2", t.n[1]); put3(s, "n3", t.n[2]);
          s += "\n";
        }
        break;
      }
    }
  }
  return s;
}

}  // namespace lfamd
