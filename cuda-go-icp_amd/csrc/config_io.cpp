// Config (.toml) parsing, PLY/TXT cloud loading and the output.toml writer: the surface the
// reference keeps in src/common.{h,cpp}, re-written small (no toml++ / tinyply dependency).
#include "config_io.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <random>
#include <sstream>
#include <stdexcept>

namespace goicp {

namespace {

std::string trim(const std::string& s)
{
	size_t a = 0, b = s.size();
	while (a < b && std::isspace((unsigned char)s[a])) a++;
	while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
	return s.substr(a, b - a);
}

// strip a trailing comment that is not inside a string
std::string strip_comment(const std::string& line)
{
	bool in_basic = false, in_literal = false;
	for (size_t i = 0; i < line.size(); i++) {
		char c = line[i];
		if (in_basic) { if (c == '\\') i++; else if (c == '"') in_basic = false; }
		else if (in_literal) { if (c == '\'') in_literal = false; }
		else if (c == '"') in_basic = true;
		else if (c == '\'') in_literal = true;
		else if (c == '#') return line.substr(0, i);
	}
	if (in_basic || in_literal) throw ConfigError("unterminated string");
	return line;
}

struct Value { std::string raw; bool is_string = false; };

std::string unescape(const std::string& s)
{
	std::string o;
	for (size_t i = 0; i < s.size(); i++) {
		if (s[i] == '\\' && i + 1 < s.size()) {
			char n = s[++i];
			switch (n) {
			case 'n': o += '\n'; break;
			case 't': o += '\t'; break;
			case '\\': o += '\\'; break;
			case '"': o += '"'; break;
			default: o += n; break;
			}
		} else o += s[i];
	}
	return o;
}

// The subset of TOML the reference's configs use: [table] / [a.b] headers, key = "string" |
// 'literal' | number | bool | [array].  Arrays (possibly multi-line) are kept raw.
std::map<std::string, Value> parse_toml_subset(std::istream& in)
{
	std::map<std::string, Value> kv;
	std::string section, line;
	int lineno = 0;
	while (std::getline(in, line)) {
		lineno++;
		std::string s;
		try { s = trim(strip_comment(line)); }
		catch (const ConfigError& e) { throw ConfigError("line " + std::to_string(lineno) + ": " + e.what()); }
		if (s.empty()) continue;
		if (s[0] == '[') {
			if (s.back() != ']') throw ConfigError("line " + std::to_string(lineno) + ": malformed table header");
			section = trim(s.substr(1, s.size() - 2));
			if (section.empty() || section[0] == '[') throw ConfigError("line " + std::to_string(lineno) + ": unsupported table header");
			continue;
		}
		size_t eq = s.find('=');
		if (eq == std::string::npos) throw ConfigError("line " + std::to_string(lineno) + ": expected key = value");
		std::string key = trim(s.substr(0, eq)), val = trim(s.substr(eq + 1));
		if (key.empty() || val.empty()) throw ConfigError("line " + std::to_string(lineno) + ": empty key or value");
		if (key.size() >= 2 && (key[0] == '"' || key[0] == '\'')) key = key.substr(1, key.size() - 2);
		Value v;
		if (val[0] == '"') {
			if (val.size() < 2 || val.back() != '"') throw ConfigError("line " + std::to_string(lineno) + ": unterminated string");
			v.raw = unescape(val.substr(1, val.size() - 2)); v.is_string = true;
		} else if (val[0] == '\'') {
			if (val.size() < 2 || val.back() != '\'') throw ConfigError("line " + std::to_string(lineno) + ": unterminated string");
			v.raw = val.substr(1, val.size() - 2); v.is_string = true;
		} else if (val[0] == '[') {
			std::string acc = val;
			int depth = 0;
			auto count = [&](const std::string& t) { for (char c : t) { if (c == '[') depth++; else if (c == ']') depth--; } };
			count(val);
			while (depth > 0 && std::getline(in, line)) { lineno++; std::string t = trim(strip_comment(line)); count(t); acc += t; }
			if (depth != 0) throw ConfigError("unterminated array for key " + key);
			v.raw = acc;
		} else {
			v.raw = val;
		}
		kv[section.empty() ? key : section + "." + key] = v;
	}
	return kv;
}

bool get_number(const std::map<std::string, Value>& kv, const std::string& k, double* out)
{
	auto it = kv.find(k);
	if (it == kv.end() || it->second.is_string) return false;
	std::string r = it->second.raw;
	r.erase(std::remove(r.begin(), r.end(), '_'), r.end());
	if (r == "inf" || r == "+inf") { *out = INFINITY; return true; }
	if (r == "-inf") { *out = -INFINITY; return true; }
	char* end = nullptr;
	double d = std::strtod(r.c_str(), &end);
	if (end == r.c_str() || *end != 0) return false;
	*out = d;
	return true;
}

float num_or(const std::map<std::string, Value>& kv, const std::string& k, float dflt)
{
	double d;
	return get_number(kv, k, &d) ? (float)d : dflt;
}

bool bool_or(const std::map<std::string, Value>& kv, const std::string& k, bool dflt)
{
	auto it = kv.find(k);
	if (it == kv.end() || it->second.is_string) return dflt;
	if (it->second.raw == "true") return true;
	if (it->second.raw == "false") return false;
	return dflt;
}

std::string str_or(const std::map<std::string, Value>& kv, const std::string& k, const std::string& dflt)
{
	auto it = kv.find(k);
	if (it == kv.end() || !it->second.is_string) return dflt;
	return it->second.raw;
}

void copy_path(char* dst, const std::string& s)
{
	std::snprintf(dst, GOICP_PATH_MAX, "%s", s.c_str());
}

}  // namespace

void load_config(const std::string& path, goicp_config* c)
{
	std::ifstream f(path);
	if (!f) throw ConfigError("Error parsing file '" + path + "': cannot open");
	std::map<std::string, Value> kv;
	try { kv = parse_toml_subset(f); }
	catch (const ConfigError& e) { throw ConfigError("Error parsing file '" + path + "': " + e.what()); }

	std::memset(c, 0, sizeof(*c));
	// defaults of the Config constructor (src/common.cpp:12-14) and of parse_toml's value_or()s
	c->mode = 1; c->trim = 0; c->subsample = 1.0f; c->mse_threshold = 1e-5f; c->resize = 1.0f;
	c->viz_theta = 0.0f; c->viz_phi = 0.4f; c->viz_spin_after_finish = 0;
	for (int k = 0; k < 3; k++) { c->rot_min[k] = -180.f; c->rot_max[k] = 180.f; c->trans_min[k] = -1.f; c->trans_max[k] = 1.f; }
	c->rot_search_depth = 12; c->trans_search_depth = 12;

	// the reference dereferences info.description unconditionally (src/common.cpp:39-40)
	auto d = kv.find("info.description");
	if (d == kv.end() || !d->second.is_string) throw ConfigError("Error parsing file '" + path + "': missing info.description");
	copy_path(c->description, d->second.raw);

	copy_path(c->target, str_or(kv, "io.target", ""));
	copy_path(c->source, str_or(kv, "io.source", ""));
	copy_path(c->output, str_or(kv, "io.output", ""));
	copy_path(c->visualization, str_or(kv, "io.visualization", ""));

	c->mode = (int)num_or(kv, "params.mode", 1);
	c->trim = bool_or(kv, "params.trim", false) ? 1 : 0;
	c->subsample = num_or(kv, "params.subsample", 1.0f);
	c->mse_threshold = num_or(kv, "params.mse_threshold", 1e-5f);
	c->resize = num_or(kv, "params.resize", 1.0f);
	c->subsample = std::min(1.0f, std::max(0.0f, c->subsample));      // src/common.cpp:63
	c->mse_threshold = std::max(1e-10f, c->mse_threshold);             // src/common.cpp:64

	c->viz_theta = num_or(kv, "visualization.theta", 0.0f);
	c->viz_phi = num_or(kv, "visualization.phi", 0.4f);
	c->viz_spin_after_finish = bool_or(kv, "visualization.spin_after_finish", false) ? 1 : 0;

	const char* ax[3] = {"x", "y", "z"};
	for (int k = 0; k < 3; k++) {
		c->rot_min[k] = num_or(kv, std::string("params.rotation.") + ax[k] + "min", -180.f);
		c->rot_max[k] = num_or(kv, std::string("params.rotation.") + ax[k] + "max", 180.f);
		c->trans_min[k] = num_or(kv, std::string("params.translation.") + ax[k] + "min", -1.f);
		c->trans_max[k] = num_or(kv, std::string("params.translation.") + ax[k] + "max", 1.f);
	}
	c->rot_search_depth = (int)num_or(kv, "params.rotation.search_depth", 12);
	c->trans_search_depth = (int)num_or(kv, "params.translation.search_depth", 12);
	for (const auto& e : kv) {
		if (e.first.rfind("params.rotation.", 0) == 0) c->has_rotation_range = 1;
		if (e.first.rfind("params.translation.", 0) == 0) c->has_translation_range = 1;
	}
}

// ------------------------------------------------------------------------------------------------
// clouds
// ------------------------------------------------------------------------------------------------
namespace {

struct Sampler {
	std::mt19937 gen;
	std::uniform_real_distribution<float> dis{0.0f, 1.0f};
	float p;
	explicit Sampler(float subsample, uint64_t seed) : p(subsample)
	{
		if (seed == 0) { std::random_device rd; gen.seed(rd()); } else gen.seed((uint32_t)seed);
	}
	bool keep() { return dis(gen) <= p; }   // src/common.cpp:122,186
};

size_t ply_type_size(const std::string& t)
{
	if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
	if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
	if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
	if (t == "double" || t == "float64") return 8;
	throw IoError("unsupported PLY property type '" + t + "'");
}

double ply_read_scalar(const unsigned char* p, const std::string& t)
{
	if (t == "float" || t == "float32") { float v; std::memcpy(&v, p, 4); return v; }
	if (t == "double" || t == "float64") { double v; std::memcpy(&v, p, 8); return v; }
	if (t == "char" || t == "int8") { int8_t v; std::memcpy(&v, p, 1); return v; }
	if (t == "uchar" || t == "uint8") { uint8_t v; std::memcpy(&v, p, 1); return v; }
	if (t == "short" || t == "int16") { int16_t v; std::memcpy(&v, p, 2); return v; }
	if (t == "ushort" || t == "uint16") { uint16_t v; std::memcpy(&v, p, 2); return v; }
	if (t == "int" || t == "int32") { int32_t v; std::memcpy(&v, p, 4); return v; }
	if (t == "uint" || t == "uint32") { uint32_t v; std::memcpy(&v, p, 4); return v; }
	throw IoError("unsupported PLY property type '" + t + "'");
}

struct PlyProp { std::string name, type; bool is_list = false; std::string count_type; };
struct PlyElem { std::string name; size_t count = 0; std::vector<PlyProp> props; };

void load_ply(const std::string& path, float subsample, float resize, uint64_t seed, std::vector<float>& out)
{
	std::ifstream f(path, std::ios::binary);
	if (!f) throw IoError("Error reading PLY file: Unable to open file: " + path);
	std::string line;
	auto getl = [&]() {
		if (!std::getline(f, line)) throw IoError("Error reading PLY file: truncated header in " + path);
		if (!line.empty() && line.back() == '\r') line.pop_back();   // CRLF headers occur in the data set
	};
	getl();
	if (trim(line) != "ply") throw IoError("Error reading PLY file: not a PLY file: " + path);
	enum { ASCII, BIN_LE } fmt = ASCII;
	std::vector<PlyElem> elems;
	while (true) {
		getl();
		std::istringstream ss(line);
		std::string tok;
		ss >> tok;
		if (tok == "end_header") break;
		if (tok == "format") {
			std::string kind; ss >> kind;
			if (kind == "ascii") fmt = ASCII;
			else if (kind == "binary_little_endian") fmt = BIN_LE;
			else throw IoError("Error reading PLY file: unsupported format '" + kind + "'");
		} else if (tok == "element") {
			PlyElem e; ss >> e.name >> e.count; elems.push_back(e);
		} else if (tok == "property") {
			if (elems.empty()) throw IoError("Error reading PLY file: property before element");
			PlyProp p; std::string t; ss >> t;
			if (t == "list") { p.is_list = true; ss >> p.count_type >> p.type >> p.name; }
			else { p.type = t; ss >> p.name; }
			elems.back().props.push_back(p);
		}
	}
	for (const PlyElem& e : elems) {
		const bool is_vertex = e.name == "vertex";
		int ix = -1, iy = -1, iz = -1;
		if (is_vertex) {
			for (size_t k = 0; k < e.props.size(); k++) {
				if (e.props[k].name == "x") ix = (int)k;
				if (e.props[k].name == "y") iy = (int)k;
				if (e.props[k].name == "z") iz = (int)k;
			}
			if (ix < 0 || iy < 0 || iz < 0)
				throw IoError("Error reading PLY file: PLY file missing 'x', 'y', or 'z' vertex properties.");
			if (e.count == 0) throw IoError("Error reading PLY file: No vertices found in the PLY file.");
		}
		Sampler smp(subsample, seed);
		const size_t cap = (size_t)((float)e.count * subsample);       // src/common.cpp:110
		size_t kept = 0;
		std::vector<double> vals(e.props.size());
		for (size_t i = 0; i < e.count; i++) {
			if (fmt == ASCII) {
				for (size_t k = 0; k < e.props.size(); k++) {
					if (e.props[k].is_list) {
						long n = 0; f >> n;
						for (long j = 0; j < n; j++) { double d; f >> d; }
					} else if (!(f >> vals[k])) throw IoError("Error reading PLY file: truncated body in " + path);
				}
			} else {
				for (size_t k = 0; k < e.props.size(); k++) {
					unsigned char buf[8];
					if (e.props[k].is_list) {
						size_t cs = ply_type_size(e.props[k].count_type), es = ply_type_size(e.props[k].type);
						f.read((char*)buf, cs);
						long n = (long)ply_read_scalar(buf, e.props[k].count_type);
						f.seekg((std::streamoff)(n * es), std::ios::cur);
					} else {
						size_t sz = ply_type_size(e.props[k].type);
						f.read((char*)buf, sz);
						if (!f) throw IoError("Error reading PLY file: truncated body in " + path);
						vals[k] = ply_read_scalar(buf, e.props[k].type);
					}
				}
			}
			if (is_vertex && kept < cap && smp.keep()) {
				out.push_back(resize * (float)vals[ix]);
				out.push_back(resize * (float)vals[iy]);
				out.push_back(resize * (float)vals[iz]);
				kept++;
			}
		}
		if (is_vertex) return;   // later elements (faces) are irrelevant
	}
	throw IoError("Error reading PLY file: No vertices found in the PLY file.");
}

void load_txt(const std::string& path, float subsample, float resize, uint64_t seed, std::vector<float>& out)
{
	std::ifstream f(path);
	if (!f.is_open()) throw IoError("Error reading TXT file: Unable to open TXT file: " + path);
	int total = 0;
	f >> total;
	if (total <= 0) throw IoError("Error reading TXT file: Invalid number of points in the TXT file: " + path);
	const size_t cap = (size_t)((float)total * subsample);             // src/common.cpp:167
	Sampler smp(subsample, seed);
	size_t kept = 0;
	for (int i = 0; i < total; i++) {
		float x, y, z;
		if (!(f >> x >> y >> z)) throw IoError("Error reading TXT file: Error reading point data from TXT file: " + path);
		if (smp.keep() && kept < cap) {                                 // src/common.cpp:186
			out.push_back(resize * x); out.push_back(resize * y); out.push_back(resize * z);
			kept++;
		}
	}
}

}  // namespace

void load_cloud(const std::string& path, float subsample, float resize, uint64_t seed, std::vector<float>& out)
{
	size_t dot = path.find_last_of('.');
	if (dot == std::string::npos) throw IoError("Filepath does not have a valid extension: " + path);
	std::string ext = path.substr(dot + 1);
	std::transform(ext.begin(), ext.end(), ext.begin(), ::tolower);
	if (ext == "ply") load_ply(path, subsample, resize, seed, out);
	else if (ext == "txt") load_txt(path, subsample, resize, seed, out);
	else throw IoError("Unsupported file extension: " + ext);
}

void write_viz_ply(const std::string& path, const float* target, size_t nt, const float* source, size_t ns)
{
	FILE* f = std::fopen(path.c_str(), "wb");
	if (!f) throw IoError("cannot write " + path);
	std::fprintf(f, "ply\nformat binary_little_endian 1.0\ncomment goicp-mi355: target (grey) + registered source (red)\n"
	                "element vertex %zu\nproperty float x\nproperty float y\nproperty float z\n"
	                "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n", nt + ns);
	auto put = [&](const float* p, size_t n, unsigned char r, unsigned char g, unsigned char b) {
		for (size_t i = 0; i < n; i++) {
			std::fwrite(p + 3 * i, sizeof(float), 3, f);
			const unsigned char c[3] = {r, g, b};
			std::fwrite(c, 1, 3, f);
		}
	};
	put(target, nt, 160, 160, 160);
	put(source, ns, 220, 40, 40);
	std::fclose(f);
}

void write_result_toml(const std::string& path, const Result& r, size_t n_source, size_t n_target, float sse_threshold, int inliers)
{
	const size_t n_used = inliers > 0 ? (size_t)inliers : n_source;      // the sums run over inlierNum points (jly_goicp.cpp:198-208)
	FILE* f = std::fopen(path.c_str(), "w");
	if (!f) throw IoError("cannot write " + path);
	std::fprintf(f, "# Go-ICP registration result (output promised by the reference configs, test/bunny_goicp.toml:12)\n");
	std::fprintf(f, "[result]\nfinished = %s\n", r.finished ? "true" : "false");
	std::fprintf(f, "sse = %.9g\nmse = %.9g\nsse_threshold = %.9g\n", r.best_sse, r.best_sse / (float)n_used, sse_threshold);
	std::fprintf(f, "rotation = [\n");
	for (int i = 0; i < 3; i++)
		std::fprintf(f, "  [%.9g, %.9g, %.9g]%s\n", r.optR[3 * i], r.optR[3 * i + 1], r.optR[3 * i + 2], i == 2 ? "" : ",");
	std::fprintf(f, "]\ntranslation = [%.9g, %.9g, %.9g]\n", r.optT[0], r.optT[1], r.optT[2]);
	std::fprintf(f, "\n[stats]\nsource_points = %zu\ninlier_points = %zu\ntarget_points = %zu\n", n_source, n_used, n_target);
	std::fprintf(f, "rotation_nodes = %lld\ntranslation_nodes = %lld\ncube_bounds = %lld\ninner_bnb_calls = %lld\n",
	             r.counters.rot_pops, r.counters.trans_pops, r.counters.cubes, r.counters.inner_calls);
	std::fprintf(f, "icp_runs = %lld\nicp_iterations = %lld\nbounds_launches = %lld\n", r.counters.icp_runs,
	             r.counters.icp_iters, r.counters.bounds_launches);
	std::fprintf(f, "dt_build_ms = %.3f\nregister_ms = %.3f\n", r.dt_build_ms, r.register_ms);
	std::fclose(f);
}

}  // namespace goicp
