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
#include "tensorflow/core/framework/tensor_util.h"
#include "tensorflow/core/framework/types.h"
#include "tensorflow/core/framework/variant.h"
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
  std::vector<std::string> input_names;   // for input("name", &tensor)
  std::vector<Tensor> outputs;            // filled by allocate_output / set_output
  Tensor output;                          // caller-provided storage for output 0, when set by the caller
  bool caller_output = false;
  Status status;
  const Tensor& input(int i) const { return inputs[i]; }
  Status input(const char* name, const Tensor** out) const {
    for (size_t i = 0; i < input_names.size(); ++i)
      if (input_names[i] == name) {
        *out = &inputs[i];
        return Status();
      }
    return errors::InvalidArgument("no input named ", name);
  }
  int num_inputs() const { return static_cast<int>(inputs.size()); }
  Status allocate_output(int i, const TensorShape& shape, Tensor** out) {
    if (i == 0 && caller_output) {
      *out = &output;
      return Status();
    }
    if (static_cast<int>(outputs.size()) <= i) outputs.resize(i + 1);
    outputs[i] = Tensor(shape);
    *out = &outputs[i];
    return Status();
  }
  void set_output(int i, const Tensor& t) {
    if (static_cast<int>(outputs.size()) <= i) outputs.resize(i + 1);
    outputs[i] = t;
  }
  void SetStatus(const Status& s) { status = s; }
  void CtxFailure(const Status& s) { status = s; }
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
#define OP_REQUIRES_VALUE(lhs, ctx, rexpr)                                   \
  do {                                                                       \
    auto v__ = (rexpr);                                                      \
    if (!v__.ok()) { (ctx)->SetStatus(v__.status()); return; }               \
    lhs = std::move(v__.value());                                            \
  } while (0)
