// Build shim (test infrastructure): just enough of the OpKernel interface to compile a
// reference CPU kernel file verbatim and call its Compute() on caller memory.
// REGISTER_KERNEL_BUILDER records a factory per element type in tfc_shim::registry().
// Not product code.
#pragma once
#include <chrono>
#include <cmath>
#include <limits>
#include <cstdint>
#include <functional>
#include <map>
#include <string>
#include <typeindex>
#include <utility>
#include <vector>

#include "tensorflow/core/framework/tensor.h"
#include "tensorflow/core/framework/types.h"
#include "tensorflow/core/lib/core/threadpool.h"
#include "tensorflow/core/platform/status.h"

namespace tensorflow {
constexpr const char* DEVICE_CPU = "CPU";
class OpKernelConstruction {
 public:
  std::map<std::string, int> int_attrs;
  Status status;
  Status GetAttr(const char* name, int* value) const {
    auto it = int_attrs.find(name);
    if (it == int_attrs.end()) return errors::InvalidArgument("no attr ", name);
    *value = it->second;
    return Status();
  }
  void SetStatus(const Status& s) { status = s; }
};
class OpKernelContext {
 public:
  std::vector<Tensor> inputs;
  Tensor output;          // caller-provided storage for output 0
  Status status;
  const Tensor& input(int i) const { return inputs[i]; }
  Status allocate_output(int, const TensorShape&, Tensor** out) {
    *out = &output;
    return Status();
  }
  void SetStatus(const Status& s) { status = s; }
  DeviceBase* device() { return &device_; }

 private:
  DeviceBase device_;
};
class OpKernel {
 public:
  explicit OpKernel(OpKernelConstruction*) {}
  virtual ~OpKernel() = default;
  virtual void Compute(OpKernelContext* context) = 0;
};
}  // namespace tensorflow

namespace tfc_shim {
using Factory = std::function<tensorflow::OpKernel*(tensorflow::OpKernelConstruction*)>;
inline std::map<std::pair<std::string, std::type_index>, Factory>& registry() {
  static std::map<std::pair<std::string, std::type_index>, Factory> r;
  return r;
}
struct KernelDef {
  std::string name;
  std::type_index type = std::type_index(typeid(void));
  KernelDef& Device(const char*) { return *this; }
  template <class T> KernelDef& TypeConstraint(const char*) {
    type = std::type_index(typeid(T));
    return *this;
  }
};
struct Registrar {
  Registrar(const KernelDef& d, Factory f) { registry()[{d.name, d.type}] = std::move(f); }
};
}  // namespace tfc_shim

namespace tensorflow {
inline tfc_shim::KernelDef Name(const char* n) {
  tfc_shim::KernelDef d;
  d.name = n;
  return d;
}
}  // namespace tensorflow

#define TFC_SHIM_CAT2(a, b) a##b
#define TFC_SHIM_CAT(a, b) TFC_SHIM_CAT2(a, b)
#define REGISTER_KERNEL_BUILDER(def, ...)                                         \
  static ::tfc_shim::Registrar TFC_SHIM_CAT(tfc_shim_registrar_, __COUNTER__)(     \
      [] { using namespace ::tensorflow; return def; }(),                          \
      [](::tensorflow::OpKernelConstruction* c) -> ::tensorflow::OpKernel* { \
        return new __VA_ARGS__(c);                                                \
      })
#define OP_REQUIRES(ctx, cond, st) \
  do { if (!(cond)) { (ctx)->SetStatus(st); return; } } while (0)
#define OP_REQUIRES_OK(ctx, expr) \
  do { ::tensorflow::Status s__ = (expr); if (!s__.ok()) { (ctx)->SetStatus(s__); return; } } while (0)
