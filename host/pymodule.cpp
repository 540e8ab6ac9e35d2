// pybind11 glue exposing the C++ tape mirror (host/neuronika.hpp) to the Python test / bench
// harness as `neuronika_amd._tape`.  Harness plumbing only: the drop-in boundary is the C ABI.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/functional.h>
#include <pybind11/stl.h>

#include "neuronika.hpp"
#include "data.hpp"

namespace py = pybind11;
using namespace neuronika;

using Arr = py::array_t<float, py::array::c_style | py::array::forcecast>;

struct Status {
    std::shared_ptr<bool> flag;
};

static Shape shape_of(const Arr& a) {
    Shape s;
    for (py::ssize_t i = 0; i < a.ndim(); ++i) s.push_back((int)a.shape(i));
    return s;
}
static py::array_t<float> to_numpy(const HipArray& h) {
    std::vector<py::ssize_t> s(h.shape().begin(), h.shape().end());
    py::array_t<float> out(s);
    h.download(out.mutable_data());
    return out;
}

PYBIND11_MODULE(_tape, m) {
    m.doc() = "C++ mirror of neuronika's Var/VarDiff tape on the HIP backend";
    py::register_exception<Panic>(m, "Panic", PyExc_RuntimeError);

    py::class_<Graph, std::shared_ptr<Graph>>(m, "Graph").def("launch", &Graph::launch);
    py::class_<Device, std::shared_ptr<Device>>(m, "Device")
        .def(py::init(&Device::create), py::arg("idx") = 0)
        .def("sync", &Device::sync)
        .def("bytes_in_use", &Device::bytes_in_use)
        .def("graph_begin", &Device::graph_begin)
        .def("graph_end", &Device::graph_end)
        .def_property_readonly("index", &Device::index)
        .def("raw", [](const Device& d) { return (uintptr_t)d.raw(); });

    py::enum_<Reduction>(m, "Reduction").value("Sum", Reduction::Sum).value("Mean", Reduction::Mean);

    py::class_<Status>(m, "Status")  // `Rc<Cell<bool>>` train/eval switch of the Dropout nodes
        .def(py::init([](bool v) { return Status{std::make_shared<bool>(v)}; }), py::arg("train") = true)
        .def("set", [](Status& s, bool v) { *s.flag = v; })
        .def("get", [](const Status& s) { return *s.flag; });

    py::class_<Var>(m, "Var")
        .def_property_readonly("shape", [](const Var& v) { return v.shape(); })
        .def("requires_grad", &Var::requires_grad)
        .def("forward", &Var::forward)
        .def("item", &Var::item)
        .def("data", [](const Var& v) { return to_numpy(*v.data); })
        .def("set_data", [](const Var& v, const Arr& a) {
            if ((size_t)a.size() != v.data->len()) panic("set_data: size mismatch");
            v.data->upload(a.data());
        })
        .def("history_len", [](const Var& v) { return v.history.len(); })
        .def("sum", &Var::sum).def("mean", &Var::mean).def("relu", &Var::relu)
        .def("__neg__", &Var::neg).def("pow", &Var::pow).def("sqrt", &Var::sqrt).def("leaky_relu", &Var::leaky_relu)
        .def("softplus", &Var::softplus).def("sigmoid", &Var::sigmoid).def("tanh", &Var::tanh).def("ln", &Var::ln)
        .def("exp", &Var::exp).def("unsqueeze", &Var::unsqueeze)
        .def("softmax", &Var::softmax).def("log_softmax", &Var::log_softmax).def("t", &Var::t)
        .def("dropout", [](const Var& v, double p, const Status& s) { return v.dropout(p, s.flag); }).def("chunks", &Var::chunks).def("cat", &Var::cat)
        .def("mse", &Var::mse).def("mae", &Var::mae).def("bce", &Var::bce).def("bce_with_logits", &Var::bce_with_logits)
        .def("kldiv", &Var::kldiv).def("nll", &Var::nll).def("stack", &Var::stack)
        .def("mv", py::overload_cast<const Var&>(&Var::mv, py::const_)).def("mv", py::overload_cast<const VarDiff&>(&Var::mv, py::const_))
        .def("vm", py::overload_cast<const Var&>(&Var::vm, py::const_)).def("vm", py::overload_cast<const VarDiff&>(&Var::vm, py::const_))
        .def("vv", py::overload_cast<const Var&>(&Var::vv, py::const_)).def("vv", py::overload_cast<const VarDiff&>(&Var::vv, py::const_)).def("pad", py::overload_cast<const std::vector<int>&, float>(&Var::pad, py::const_), py::arg("padding"), py::arg("value") = 0.f)
        .def("pad", py::overload_cast<const std::vector<int>&, PaddingMode>(&Var::pad, py::const_))
        .def("mm", py::overload_cast<const Var&>(&Var::mm, py::const_))
        .def("mm", py::overload_cast<const VarDiff&>(&Var::mm, py::const_))
        .def("mm_t", py::overload_cast<const Var&>(&Var::mm_t, py::const_))
        .def("mm_t", py::overload_cast<const VarDiff&>(&Var::mm_t, py::const_))
        .def("heads_attention", [](const Var& q, const Var& k, const Var& v, int B, int S, int H, int dh, float scale, double p, const Status& s) {
            return q.heads_attention(k, v, B, S, H, dh, scale, p, s.flag); })
        .def_static("attention_core_supported", &Var::attention_core_supported)
        .def("convolution", &Var::convolution)
        .def("__add__", [](const Var& a, const Var& b) { return a + b; })
        .def("__add__", [](const Var& a, const VarDiff& b) { return a + b; })
        .def("__add__", [](const Var& a, float b) { return a + b; })
        .def("__sub__", [](const Var& a, const Var& b) { return a - b; })
        .def("__sub__", [](const Var& a, const VarDiff& b) { return a - b; })
        .def("__sub__", [](const Var& a, float b) { return a - b; })
        .def("__mul__", [](const Var& a, const Var& b) { return a * b; })
        .def("__mul__", [](const Var& a, const VarDiff& b) { return a * b; })
        .def("__mul__", [](const Var& a, float b) { return a * b; })
        .def("__truediv__", [](const Var& a, const Var& b) { return a / b; })
        .def("__truediv__", [](const Var& a, const VarDiff& b) { return a / b; })
        .def("__truediv__", [](const Var& a, float b) { return a / b; });

    py::class_<VarDiff>(m, "VarDiff")
        .def_property_readonly("shape", [](const VarDiff& v) { return v.shape(); })
        .def("forward", &VarDiff::forward)
        .def("backward", [](const VarDiff& v, float seed) { v.backward(seed); }, py::arg("seed") = 1.f)
        .def("backward_from", [](const VarDiff& v, const Var& seed) { v.backward_from(seed); }, py::arg("seed"))
        .def("backward_sync", [](const VarDiff& v, float seed, dp::GradientSync& s) { v.backward(seed, &s); })
        .def("zero_grad", &VarDiff::zero_grad)
        .def("no_grad", &VarDiff::no_grad)
        .def("with_grad", &VarDiff::with_grad)
        .def("item", &VarDiff::item)
        .def("data", [](const VarDiff& v) { return to_numpy(*v.var.data); })
        .def("set_data", [](const VarDiff& v, const Arr& a) {
            if ((size_t)a.size() != v.var.data->len()) panic("set_data: size mismatch");
            v.var.data->upload(a.data());
        })
        .def("grad", [](const VarDiff& v) { return to_numpy(v.grad->borrow()); })
        .def("set_grad", [](const VarDiff& v, const Arr& a) {
            if ((size_t)a.size() != v.grad->borrow().len()) panic("set_grad: size mismatch");
            v.grad->borrow().upload(a.data());
        })
        .def("history_len", [](const VarDiff& v) { return v.history.len(); })
        .def("forward_history_len", [](const VarDiff& v) { return v.var.history.len(); })
        .def("sum", &VarDiff::sum).def("mean", &VarDiff::mean)
        // Python has no rvalues: a receiver nobody else references (`lin.forward(x).relu()`: the only reference is the call's own)
        // is the temporary Rust's `relu(self)` consumes; a named variable stays observable and gets the plain ReLU node
        .def("relu", [](py::handle self) {
            VarDiff& v = self.cast<VarDiff&>();
            return Py_REFCNT(self.ptr()) <= 1 ? std::move(v).relu() : v.relu();
        })
        .def("__neg__", &VarDiff::neg).def("pow", &VarDiff::pow).def("sqrt", &VarDiff::sqrt).def("leaky_relu", &VarDiff::leaky_relu)
        .def("softplus", &VarDiff::softplus).def("sigmoid", &VarDiff::sigmoid).def("tanh", &VarDiff::tanh).def("ln", &VarDiff::ln)
        .def("exp", &VarDiff::exp).def("unsqueeze", &VarDiff::unsqueeze)
        .def("softmax", &VarDiff::softmax).def("log_softmax", &VarDiff::log_softmax).def("t", &VarDiff::t)
        .def("dropout", [](const VarDiff& v, double p, const Status& s) { return v.dropout(p, s.flag); }).def("chunks", &VarDiff::chunks).def("cat", &VarDiff::cat)
        .def("mse", &VarDiff::mse).def("mae", &VarDiff::mae).def("bce", &VarDiff::bce).def("bce_with_logits", &VarDiff::bce_with_logits)
        .def("kldiv", &VarDiff::kldiv).def("nll", &VarDiff::nll).def("stack", &VarDiff::stack)
        .def("mv", py::overload_cast<const Var&>(&VarDiff::mv, py::const_)).def("mv", py::overload_cast<const VarDiff&>(&VarDiff::mv, py::const_))
        .def("vm", py::overload_cast<const Var&>(&VarDiff::vm, py::const_)).def("vm", py::overload_cast<const VarDiff&>(&VarDiff::vm, py::const_))
        .def("vv", py::overload_cast<const Var&>(&VarDiff::vv, py::const_)).def("vv", py::overload_cast<const VarDiff&>(&VarDiff::vv, py::const_)).def("pad", py::overload_cast<const std::vector<int>&, float>(&VarDiff::pad, py::const_), py::arg("padding"), py::arg("value") = 0.f)
        .def("pad", py::overload_cast<const std::vector<int>&, PaddingMode>(&VarDiff::pad, py::const_))
        .def("mm", py::overload_cast<const Var&>(&VarDiff::mm, py::const_))
        .def("mm", py::overload_cast<const VarDiff&>(&VarDiff::mm, py::const_))
        .def("mm_t", py::overload_cast<const Var&>(&VarDiff::mm_t, py::const_))
        .def("mm_t", py::overload_cast<const VarDiff&>(&VarDiff::mm_t, py::const_))
        .def("heads_attention", [](const VarDiff& q, const VarDiff& k, const VarDiff& v, int B, int S, int H, int dh, float scale, double p, const Status& s) {
            return q.heads_attention(k, v, B, S, H, dh, scale, p, s.flag); })
        .def("convolution", py::overload_cast<const Var&, const std::vector<int>&, const std::vector<int>&, int>(&VarDiff::convolution, py::const_))
        .def("convolution", py::overload_cast<const VarDiff&, const std::vector<int>&, const std::vector<int>&, int>(&VarDiff::convolution, py::const_))
        .def("__add__", [](const VarDiff& a, const Var& b) { return a + b; })
        .def("__add__", [](const VarDiff& a, const VarDiff& b) { return a + b; })
        .def("__add__", [](const VarDiff& a, float b) { return a + b; })
        .def("__sub__", [](const VarDiff& a, const Var& b) { return a - b; })
        .def("__sub__", [](const VarDiff& a, const VarDiff& b) { return a - b; })
        .def("__sub__", [](const VarDiff& a, float b) { return a - b; })
        .def("__mul__", [](const VarDiff& a, const Var& b) { return a * b; })
        .def("__mul__", [](const VarDiff& a, const VarDiff& b) { return a * b; })
        .def("__mul__", [](const VarDiff& a, float b) { return a * b; })
        .def("__truediv__", [](const VarDiff& a, const Var& b) { return a / b; })
        .def("__truediv__", [](const VarDiff& a, const VarDiff& b) { return a / b; })
        .def("__truediv__", [](const VarDiff& a, float b) { return a / b; });

    m.def("from_ndarray", [](DevicePtr dev, const Arr& a) { return from_host(std::move(dev), shape_of(a), a.data()); });
    m.def("zeros", &zeros);
    m.def("ones", &ones);
    m.def("full", &full);
    m.def("rand", &neuronika::rand);
    m.def("manual_seed", &neuronika::manual_seed, py::arg("seed"));
    m.def("eye", &eye); m.def("linspace", &linspace); m.def("logspace", &logspace); m.def("geomspace", &geomspace);
    m.def("range", &neuronika::range);

    {   // neuronika-data mirror + device input pipeline
        namespace nd = neuronika::data;
        py::module_ dm = m.def_submodule("data");
        auto to_np = [](const nd::HostArray& a) {
            std::vector<py::ssize_t> shape(a.shape().begin(), a.shape().end());
            Arr out(shape);
            if (a.len()) std::memcpy(out.mutable_data(), a.ptr(), a.len() * sizeof(float));
            return out;
        };
        auto from_np = [](const Arr& a) { return nd::HostArray(shape_of(a), a.data()); };
        dm.def("batch_ranges", [](size_t len, size_t size, bool drop_last) {
            std::vector<std::pair<size_t, size_t>> out;
            for (const auto& r : nd::batch_ranges(len, size, drop_last)) out.emplace_back(r.start, r.rows);
            return out;
        });
        dm.def("kfold_ids", [](size_t len, size_t k) {
            std::vector<std::pair<std::vector<size_t>, std::vector<size_t>>> out;
            for (const auto& f : nd::kfold_ids(len, k)) out.emplace_back(f.train_ids, f.test_ids);
            return out;
        });
        dm.def("shuffle_permutation", &nd::shuffle_permutation);
        py::class_<nd::Dataset>(dm, "Dataset")
            .def(py::init([from_np](const Arr& a) { return new nd::Dataset(from_np(a)); }))
            .def("records", [to_np](const nd::Dataset& d) { return to_np(d.records()); })
            .def("pinned", [](const nd::Dataset& d) { return d.records().pinned(); })
            .def("__len__", &nd::Dataset::len).def("is_empty", &nd::Dataset::is_empty)
            .def("kfold", &nd::Dataset::kfold)
            .def("batch", [to_np](const nd::Dataset& d, size_t n, bool drop_last) {
                std::vector<Arr> out;
                for (const auto& b : d.batch(n, drop_last)) out.push_back(to_np(b));
                return out;
            }, py::arg("size"), py::arg("drop_last") = false)
            .def("split", &nd::Dataset::split)
            .def("shuffle_with_seed", [](nd::Dataset& d, uint64_t seed) { d.shuffle_with_seed(seed); });
        py::class_<nd::LabeledDataset>(dm, "LabeledDataset")
            .def(py::init([from_np](const Arr& r, const Arr& l) { return new nd::LabeledDataset(from_np(r), from_np(l)); }))
            .def("records", [to_np](const nd::LabeledDataset& d) { return to_np(d.records()); })
            .def("labels", [to_np](const nd::LabeledDataset& d) { return to_np(d.labels()); })
            .def("__len__", &nd::LabeledDataset::len).def("is_empty", &nd::LabeledDataset::is_empty)
            .def("kfold", &nd::LabeledDataset::kfold)
            .def("batch", [to_np](const nd::LabeledDataset& d, size_t n, bool drop_last) {
                std::vector<std::pair<Arr, Arr>> out;
                for (const auto& b : d.batch(n, drop_last)) out.emplace_back(to_np(b.first), to_np(b.second));
                return out;
            }, py::arg("size"), py::arg("drop_last") = false)
            .def("split", &nd::LabeledDataset::split)
            .def("shuffle_with_seed", [](nd::LabeledDataset& d, uint64_t seed) { d.shuffle_with_seed(seed); });
        py::class_<nd::DataLoader>(dm, "DataLoader")
            .def(py::init<>())
            .def("without_headers", [](nd::DataLoader& l) { l.without_headers(); return l; })
            .def("with_delimiter", [](nd::DataLoader& l, char d) { l.with_delimiter(d); return l; })
            .def("with_labels", &nd::DataLoader::with_labels)
            .def("from_csv", &nd::DataLoader::from_csv)
            .def("from_string", &nd::DataLoader::from_string);
        py::class_<nd::LabeledDataLoader>(dm, "LabeledDataLoader")
            .def("without_headers", [](nd::LabeledDataLoader& l) { l.without_headers(); return l; })
            .def("with_delimiter", [](nd::LabeledDataLoader& l, char d) { l.with_delimiter(d); return l; })
            .def("from_csv", &nd::LabeledDataLoader::from_csv)
            .def("from_string", &nd::LabeledDataLoader::from_string);
        py::class_<nd::DeviceLoader>(dm, "DeviceLoader")
            .def(py::init<DevicePtr, const nd::LabeledDataset&, size_t, bool>(), py::arg("dev"), py::arg("dataset"),
                 py::arg("batch_size"), py::arg("drop_last") = true)
            .def(py::init<DevicePtr, const nd::Dataset&, size_t, bool>(), py::arg("dev"), py::arg("dataset"),
                 py::arg("batch_size"), py::arg("drop_last") = true)
            .def("batches", &nd::DeviceLoader::batches)
            .def("next_into", [](nd::DeviceLoader& l, const Var& x) { return l.next_into(x, nullptr); })
            .def("next_into", [](nd::DeviceLoader& l, const Var& x, const Var& y) { return l.next_into(x, &y); })
            .def("next", [](nd::DeviceLoader& l) {
                auto b = l.next();
                return py::make_tuple(b.rows, b.rows ? py::cast(b.x) : py::none(), (b.rows && b.y.data) ? py::cast(b.y) : py::none());
            });
    }

    py::module_ sd = m.def_submodule("serde");
    sd.def("to_json", py::overload_cast<const Var&>(&serde::to_json));
    sd.def("to_json", py::overload_cast<const VarDiff&>(&serde::to_json));
    sd.def("to_json", py::overload_cast<const nn::Linear&>(&serde::to_json));
    sd.def("var_from_json", py::overload_cast<DevicePtr, const std::string&>(&serde::var_from_json));
    sd.def("vardiff_from_json", py::overload_cast<DevicePtr, const std::string&>(&serde::vardiff_from_json));
    sd.def("linear_from_json", py::overload_cast<DevicePtr, const std::string&>(&serde::linear_from_json));

    py::module_ nn = m.def_submodule("nn");
    nn.def("set_relu_peephole", &nn::set_relu_peephole, py::arg("on"));
    py::class_<nn::Linear>(nn, "Linear")
        .def(py::init<DevicePtr, int, int, uint64_t>(), py::arg("dev"), py::arg("in_features"), py::arg("out_features"), py::arg("seed") = 0)
        .def(py::init<VarDiff, VarDiff>())
        .def_readonly("weight", &nn::Linear::weight)
        .def_readonly("bias", &nn::Linear::bias)
        .def_readwrite("fused", &nn::Linear::fused)
        .def("forward", py::overload_cast<const Var&>(&nn::Linear::forward, py::const_))
        .def("forward", py::overload_cast<const VarDiff&>(&nn::Linear::forward, py::const_))
        .def("forward_relu", py::overload_cast<const Var&>(&nn::Linear::forward_relu, py::const_))
        .def("forward_relu", py::overload_cast<const VarDiff&>(&nn::Linear::forward_relu, py::const_));
    {
        py::module_ im = nn.def_submodule("init");
        im.def("calculate_gain", &nn::init::calculate_gain);
        im.def("calculate_fan_in_fan_out", &nn::init::calculate_fan_in_fan_out);
        im.def("constant", &nn::init::constant); im.def("zeros", &nn::init::zeros); im.def("ones", &nn::init::ones);
        im.def("eye", &nn::init::eye); im.def("dirac", &nn::init::dirac);
        im.def("uniform", &nn::init::uniform, py::arg("param"), py::arg("low"), py::arg("high"), py::arg("seed") = 0);
        im.def("normal", &nn::init::normal, py::arg("param"), py::arg("mean"), py::arg("std"), py::arg("seed") = 0);
        im.def("xavier_uniform", &nn::init::xavier_uniform, py::arg("param"), py::arg("gain"), py::arg("seed") = 0);
        im.def("xavier_normal", &nn::init::xavier_normal, py::arg("param"), py::arg("gain"), py::arg("seed") = 0);
    }
    using State = std::pair<VarDiff, VarDiff>;
    py::class_<nn::LSTMCell>(nn, "LSTMCell")
        .def(py::init<DevicePtr, int, int, uint64_t>(), py::arg("dev"), py::arg("input_size"), py::arg("hidden_size"), py::arg("seed") = 0)
        .def_readonly("weight_ih", &nn::LSTMCell::weight_ih).def_readonly("weight_hh", &nn::LSTMCell::weight_hh)
        .def_readonly("bias_ih", &nn::LSTMCell::bias_ih).def_readonly("bias_hh", &nn::LSTMCell::bias_hh)
        .def("forward", py::overload_cast<const State&, const Var&>(&nn::LSTMCell::forward, py::const_))
        .def("forward", py::overload_cast<const State&, const VarDiff&>(&nn::LSTMCell::forward, py::const_));
    py::class_<nn::GRUCell>(nn, "GRUCell")
        .def(py::init<DevicePtr, int, int, uint64_t>(), py::arg("dev"), py::arg("input_size"), py::arg("hidden_size"), py::arg("seed") = 0)
        .def_readonly("weight_ih", &nn::GRUCell::weight_ih).def_readonly("weight_hh", &nn::GRUCell::weight_hh)
        .def_readonly("bias_ih", &nn::GRUCell::bias_ih).def_readonly("bias_hh", &nn::GRUCell::bias_hh)
        .def("forward", py::overload_cast<const VarDiff&, const Var&>(&nn::GRUCell::forward, py::const_))
        .def("forward", py::overload_cast<const VarDiff&, const VarDiff&>(&nn::GRUCell::forward, py::const_));
    py::class_<PaddingMode>(m, "PaddingMode")
        .def_static("zero", &PaddingMode::zero).def_static("constant", &PaddingMode::constant)
        .def_static("reflective", &PaddingMode::reflective).def_static("replicative", &PaddingMode::replicative);
    py::class_<nn::ConvNd>(nn, "ConvNd")
        .def_readonly("weight", &nn::ConvNd::weight)
        .def_readonly("bias", &nn::ConvNd::bias)
        .def_readwrite("fused", &nn::ConvNd::fused)
        .def_readwrite("fold_padding", &nn::ConvNd::fold_padding)
        .def_readonly("groups", &nn::ConvNd::groups)
        .def_readonly("padding", &nn::ConvNd::padding).def_readonly("stride", &nn::ConvNd::stride).def_readonly("dilation", &nn::ConvNd::dilation)
        .def("forward", py::overload_cast<const Var&>(&nn::ConvNd::forward, py::const_))
        .def("forward", py::overload_cast<const VarDiff&>(&nn::ConvNd::forward, py::const_));
    // argument order of the reference's `new` (neuronika-nn/src/lib.rs:671-679, 762-770, 857-865) + seed; GroupedConv*: + groups
    py::class_<nn::Conv1d, nn::ConvNd>(nn, "Conv1d")
        .def(py::init<DevicePtr, int, int, int, int, PaddingMode, int, int, uint64_t>(), py::arg("dev"), py::arg("in_channels"),
             py::arg("out_channels"), py::arg("kernel_size"), py::arg("padding"), py::arg("padding_mode"), py::arg("stride"),
             py::arg("dilation"), py::arg("seed") = 0);
    py::class_<nn::Conv2d, nn::ConvNd>(nn, "Conv2d")
        .def(py::init<DevicePtr, int, int, std::vector<int>, std::vector<int>, PaddingMode, std::vector<int>, std::vector<int>, uint64_t>(),
             py::arg("dev"), py::arg("in_channels"), py::arg("out_channels"), py::arg("kernel_size"), py::arg("padding"),
             py::arg("padding_mode"), py::arg("stride"), py::arg("dilation"), py::arg("seed") = 0);
    py::class_<nn::Conv3d, nn::ConvNd>(nn, "Conv3d")
        .def(py::init<DevicePtr, int, int, std::vector<int>, std::vector<int>, PaddingMode, std::vector<int>, std::vector<int>, uint64_t>(),
             py::arg("dev"), py::arg("in_channels"), py::arg("out_channels"), py::arg("kernel_size"), py::arg("padding"),
             py::arg("padding_mode"), py::arg("stride"), py::arg("dilation"), py::arg("seed") = 0);
    py::class_<nn::GroupedConv1d, nn::ConvNd>(nn, "GroupedConv1d")
        .def(py::init<DevicePtr, int, int, int, int, PaddingMode, int, int, int, uint64_t>(), py::arg("dev"), py::arg("in_channels"),
             py::arg("out_channels"), py::arg("kernel_size"), py::arg("padding"), py::arg("padding_mode"), py::arg("stride"),
             py::arg("dilation"), py::arg("groups"), py::arg("seed") = 0);
    py::class_<nn::GroupedConv2d, nn::ConvNd>(nn, "GroupedConv2d")
        .def(py::init<DevicePtr, int, int, std::vector<int>, std::vector<int>, PaddingMode, std::vector<int>, std::vector<int>, int, uint64_t>(),
             py::arg("dev"), py::arg("in_channels"), py::arg("out_channels"), py::arg("kernel_size"), py::arg("padding"),
             py::arg("padding_mode"), py::arg("stride"), py::arg("dilation"), py::arg("groups"), py::arg("seed") = 0);
    py::class_<nn::GroupedConv3d, nn::ConvNd>(nn, "GroupedConv3d")
        .def(py::init<DevicePtr, int, int, std::vector<int>, std::vector<int>, PaddingMode, std::vector<int>, std::vector<int>, int, uint64_t>(),
             py::arg("dev"), py::arg("in_channels"), py::arg("out_channels"), py::arg("kernel_size"), py::arg("padding"),
             py::arg("padding_mode"), py::arg("stride"), py::arg("dilation"), py::arg("groups"), py::arg("seed") = 0);
    py::class_<nn::Dropout>(nn, "Dropout")
        .def(py::init<double>())
        .def("train", &nn::Dropout::train)
        .def("eval", &nn::Dropout::eval)
        .def("forward", &nn::Dropout::forward);
    py::class_<nn::MultiheadAttention>(nn, "MultiheadAttention")
        .def(py::init<DevicePtr, int, int, double, uint64_t>(), py::arg("dev"), py::arg("d_model"), py::arg("heads"),
             py::arg("p") = 0.0, py::arg("seed") = 0)
        .def_readwrite("q", &nn::MultiheadAttention::q)   // public, assignable members in the C++ mirror as well
        .def_readwrite("k", &nn::MultiheadAttention::k)
        .def_readwrite("v", &nn::MultiheadAttention::v)
        .def_readwrite("o", &nn::MultiheadAttention::o)
        .def_readonly("drop", &nn::MultiheadAttention::drop)
        .def_readwrite("fused", &nn::MultiheadAttention::fused)
        .def_readwrite("strided_heads", &nn::MultiheadAttention::strided_heads)
        .def_readwrite("fused_core", &nn::MultiheadAttention::fused_core)
        .def_readwrite("packed_qkv", &nn::MultiheadAttention::packed_qkv)
        .def("forward", &nn::MultiheadAttention::forward);

    py::module_ optim = m.def_submodule("optim");
    {
        namespace ls = optim::lr_scheduler;
        py::module_ lm = optim.def_submodule("lr_scheduler");
        py::class_<ls::LRScheduler>(lm, "LRScheduler")
            .def("step", &ls::LRScheduler::step)
            .def("get_last_lr", &ls::LRScheduler::get_last_lr)
            .def("get_current_lr", &ls::LRScheduler::get_current_lr)
            .def("get_current_epoch", &ls::LRScheduler::get_current_epoch)
            .def("set_current_epoch", &ls::LRScheduler::set_current_epoch);
        py::class_<ls::StepLR, ls::LRScheduler>(lm, "StepLR")
            .def(py::init<optim::Optimizer&, size_t, float>(), py::keep_alive<1, 2>()).def("set_gamma", &ls::StepLR::set_gamma);
        py::class_<ls::MultiStepLR, ls::LRScheduler>(lm, "MultiStepLR")
            .def(py::init<optim::Optimizer&, std::vector<size_t>, float>(), py::keep_alive<1, 2>())
            .def("set_milestones", &ls::MultiStepLR::set_milestones);
        py::class_<ls::ExponentialLR, ls::LRScheduler>(lm, "ExponentialLR")
            .def(py::init<optim::Optimizer&, float>(), py::keep_alive<1, 2>()).def("set_gamma", &ls::ExponentialLR::set_gamma);
        py::class_<ls::LambdaLR, ls::LRScheduler>(lm, "LambdaLR")
            .def(py::init<optim::Optimizer&, std::function<float(size_t)>>(), py::keep_alive<1, 2>());
        py::class_<ls::MultiplicativeLR, ls::LRScheduler>(lm, "MultiplicativeLR")
            .def(py::init<optim::Optimizer&, std::function<float(size_t)>>(), py::keep_alive<1, 2>());
    }
    py::class_<optim::Optimizer>(optim, "Optimizer")
        .def("register", &optim::Optimizer::register_param)
        .def("step", &optim::Optimizer::step)
        .def("zero_grad", &optim::Optimizer::zero_grad)
        .def("get_lr", &optim::Optimizer::get_lr)
        .def("set_lr", &optim::Optimizer::set_lr);
    py::class_<optim::SGD, optim::Optimizer>(optim, "SGD")
        .def(py::init([](float lr, float l1, float l2, float momentum, float dampening, bool nesterov) {
                 return new optim::SGD(lr, optim::Penalty{l1, l2}, momentum, dampening, nesterov);
             }),
             py::arg("lr"), py::arg("l1") = 0.f, py::arg("l2") = 0.f, py::arg("momentum") = 0.f, py::arg("dampening") = 0.f,
             py::arg("nesterov") = false);
    py::class_<optim::Adam, optim::Optimizer>(optim, "Adam")
        .def(py::init([](float lr, float beta1, float beta2, float eps, float l1, float l2, bool amsgrad) {
                 return new optim::Adam(lr, beta1, beta2, eps, optim::Penalty{l1, l2}, amsgrad);
             }),
             py::arg("lr"), py::arg("beta1") = 0.9f, py::arg("beta2") = 0.999f, py::arg("eps") = 1e-8f, py::arg("l1") = 0.f,
             py::arg("l2") = 0.f, py::arg("amsgrad") = false);
    py::class_<optim::Adagrad, optim::Optimizer>(optim, "Adagrad")
        .def(py::init([](float lr, float lr_decay, float eps, float l1, float l2) {
                 return new optim::Adagrad(lr, lr_decay, eps, optim::Penalty{l1, l2});
             }),
             py::arg("lr"), py::arg("lr_decay") = 0.f, py::arg("eps") = 1e-10f, py::arg("l1") = 0.f, py::arg("l2") = 0.f);
    py::class_<optim::RMSProp, optim::Optimizer>(optim, "RMSProp")
        .def(py::init([](float lr, float alpha, float eps, float momentum, bool centered, float l1, float l2) {
                 return new optim::RMSProp(lr, alpha, eps, momentum, centered, optim::Penalty{l1, l2});
             }),
             py::arg("lr"), py::arg("alpha") = 0.99f, py::arg("eps") = 1e-8f, py::arg("momentum") = 0.f,
             py::arg("centered") = false, py::arg("l1") = 0.f, py::arg("l2") = 0.f);

    py::module_ dpm = m.def_submodule("dp");
    py::class_<dp::Communicator, std::shared_ptr<dp::Communicator>>(dpm, "Communicator")
        .def_static("unique_id", []() { return py::bytes(dp::Communicator::unique_id()); })
        .def(py::init([](DevicePtr dev, int nranks, int rank, py::bytes id) {
            return std::make_shared<dp::Communicator>(std::move(dev), nranks, rank, std::string(id));
        }))
        .def_static("replicas", &dp::Communicator::replicas, py::arg("device"), py::arg("nranks"), py::arg("channels") = 0, py::arg("gbps") = 0.0)
        .def("raw", [](const dp::Communicator& c) { return (uintptr_t)c.raw(); })  // nk_comm* for the raw C ABI (bench diagnostics)
        .def_property_readonly("rank", &dp::Communicator::rank)
        .def_property_readonly("size", &dp::Communicator::size);
    py::class_<dp::GradientSync>(dpm, "GradientSync")
        .def(py::init<std::shared_ptr<dp::Communicator>, const std::vector<VarDiff>&, size_t>(), py::arg("comm"), py::arg("params"),
             py::arg("small_elems") = (size_t)65536)
        .def("join", &dp::GradientSync::join)
        .def("bytes_per_step", &dp::GradientSync::bytes_per_step)
        .def("set_force_exchange", &dp::GradientSync::set_force_exchange)
        .def("set_busy_slots", &dp::GradientSync::set_busy_slots, py::arg("n"))
        .def("busy_slots", &dp::GradientSync::busy_slots)
        .def("set_parts", [](dp::GradientSync& s, const std::string& mode) {
            if (mode == "all") s.set_parts(dp::GradientSync::Parts::All);
            else if (mode == "last") s.set_parts(dp::GradientSync::Parts::LastOnly);
            else if (mode == "none") s.set_parts(dp::GradientSync::Parts::None);
            else throw std::invalid_argument("set_parts: all | last | none");
        })
        .def("exchanges_issued", &dp::GradientSync::exchanges_issued)
        .def("elements_exchanged", &dp::GradientSync::elements_exchanged);
    dpm.def("all_reduce_gradients", &dp::all_reduce_gradients);
}
