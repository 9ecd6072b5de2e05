// mini-yaml: a stand-in for <yaml-cpp/yaml.h> — yaml-cpp is not installed in the build container.
// Enough of YAML::Node for the reference's include/read_configs.h to compile UNCHANGED and to read AirSLAM's own config
// files (configs/**/*.yaml: nested block maps by indentation, scalar leaves, `#` comments, quoted strings, `- item` lists).
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace YAML {
class Node {
 public:
  Node() : d_(std::make_shared<Data>()) {}
  const Node operator[](const std::string& key) const {
    auto it = d_->map.find(key);
    if (it == d_->map.end()) return Node();              // yaml-cpp: an undefined node; as<T>() on it throws
    return it->second;
  }
  const Node operator[](const char* key) const { return (*this)[std::string(key)]; }
  const Node operator[](size_t i) const { return i < d_->seq.size() ? d_->seq[i] : Node(); }
  const Node operator[](int i) const { return (*this)[(size_t)i]; }
  size_t size() const { return d_->seq.empty() ? d_->map.size() : d_->seq.size(); }
  bool IsDefined() const { return d_->defined; }
  explicit operator bool() const { return d_->defined; }
  template <class T>
  T as() const {
    if (!d_->defined || !d_->map.empty() || !d_->seq.empty()) throw std::runtime_error("mini-yaml: bad conversion");
    return conv<T>(d_->scalar);
  }

 private:
  struct Data {
    bool defined = false;
    std::string scalar;
    std::map<std::string, Node> map;
    std::vector<Node> seq;
  };
  std::shared_ptr<Data> d_;
  template <class T>
  static T conv(const std::string& s) {
    std::istringstream ss(s);
    T v;
    ss >> v;
    if (ss.fail()) throw std::runtime_error("mini-yaml: bad conversion of '" + s + "'");
    return v;
  }
  friend Node LoadFile(const std::string&);
  friend Node Load(const std::string&);
  static std::string strip(std::string s) {
    bool q = false;
    for (size_t i = 0; i < s.size(); ++i) {              // cut a trailing comment (outside quotes)
      if (s[i] == '"' || s[i] == '\'') q = !q;
      if (s[i] == '#' && !q && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) { s.resize(i); break; }
    }
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    s = s.substr(a, b - a + 1);
    if (s.size() >= 2 && (s.front() == '"' || s.front() == '\'') && s.back() == s.front()) s = s.substr(1, s.size() - 2);
    return s;
  }
  static Node parse(std::istream& in) {
    Node root;
    root.d_->defined = true;
    std::vector<std::pair<int, Node>> stack{{-1, root}};
    std::string line;
    while (std::getline(in, line)) {
      if (strip(line).empty() || strip(line) == "%YAML:1.0" || strip(line) == "---") continue;
      const int indent = (int)line.find_first_not_of(' ');
      while (stack.size() > 1 && stack.back().first >= indent) stack.pop_back();
      Node parent = stack.back().second;
      std::string body = line.substr(indent);
      if (body.rfind("- ", 0) == 0) {                    // sequence item
        Node item;
        item.d_->defined = true;
        item.d_->scalar = strip(body.substr(2));
        parent.d_->seq.push_back(item);
        continue;
      }
      const size_t colon = body.find(':');
      if (colon == std::string::npos) continue;
      Node child;
      child.d_->defined = true;
      child.d_->scalar = strip(body.substr(colon + 1));
      parent.d_->map[strip(body.substr(0, colon))] = child;
      if (child.d_->scalar.empty()) stack.push_back({indent, child});
    }
    return root;
  }
};
template <>
inline std::string Node::conv<std::string>(const std::string& s) { return s; }

inline Node LoadFile(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("mini-yaml: cannot open " + path);
  return Node::parse(f);
}
inline Node Load(const std::string& text) {
  std::istringstream f(text);
  return Node::parse(f);
}
}  // namespace YAML
