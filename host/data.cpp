// See data.hpp.  Reference: neuronika-data/src/lib.rs (line numbers cited per function).
#include "data.hpp"

#include <algorithm>
#include <charconv>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>

namespace neuronika {
namespace data {

namespace {
[[noreturn]] void fail(const std::string& m) { throw Panic(m); }
void check(int rc) {
    if (rc != 0) fail(std::string("neuronika_hip: ") + nk_last_error());
}
size_t numel_of(const Shape& s) {
    size_t n = 1;
    for (int d : s) n *= (size_t)d;
    return n;
}
Shape stacked_shape(size_t rows, const Shape& record) {  // lib.rs:16-23
    Shape s{(int)rows};
    s.insert(s.end(), record.begin(), record.end());
    return s;
}
}  // namespace

// ------------------------------------------------------------------------------------ HostArray
HostArray::HostArray(Shape shape) : shape_(std::move(shape)), len_(numel_of(shape_)) {
    if (len_ == 0) return;
    void* p = nullptr;
    int ndev = 0;
    if (nk_device_count(&ndev) == 0 && ndev > 0 && nk_host_alloc(len_ * sizeof(float), &p) == 0 && p) {
        pinned_ = true;
        data_ = std::shared_ptr<float>((float*)p, [](float* q) { (void)nk_host_free(q); });
    } else {
        data_ = std::shared_ptr<float>(new float[len_], std::default_delete<float[]>());
    }
}
HostArray::HostArray(Shape shape, const float* src) : HostArray(std::move(shape)) {
    if (len_) std::memcpy(ptr(), src, len_ * sizeof(float));
}
HostArray HostArray::select_rows(const std::vector<size_t>& ids) const {
    Shape s = shape_;
    s[0] = (int)ids.size();
    HostArray out(s);
    const size_t rl = row_len();
    for (size_t i = 0; i < ids.size(); ++i) {
        if (ids[i] >= rows()) fail("select: index out of bounds");
        std::memcpy(out.ptr() + i * rl, ptr() + ids[i] * rl, rl * sizeof(float));
    }
    return out;
}
HostArray HostArray::slice_rows(size_t start, size_t count) const {
    if (start + count > rows()) fail("slice: out of bounds");
    Shape s = shape_;
    s[0] = (int)count;
    return HostArray(s, ptr() + start * row_len());
}

// ------------------------------------------------------------------------------------ index logic
std::vector<BatchRange> batch_ranges(size_t len, size_t size, bool drop_last) {
    if (size == 0) fail("chunk size must be greater than zero");  // ndarray's axis_chunks_iter assertion
    std::vector<BatchRange> out;
    for (size_t s = 0; s < len; s += size) out.push_back({s, std::min(size, len - s)});
    if (drop_last && out.size() > 1 && out.back().rows != out.front().rows) out.pop_back();
    return out;
}
std::vector<Fold> kfold_ids(size_t len, size_t k) {
    if (k < 2) fail("error: folds must be > 2.");  // lib.rs:694 (message as in the reference)
    if (len == 0) fail("no record provided");
    const size_t step = 1 + (len - 1) / k;
    std::vector<Fold> folds(k);
    for (size_t i = 0; i < k; ++i) {
        const size_t start = std::min(len, step * i), stop = std::min(len, start + step);
        for (size_t r = 0; r < start; ++r) folds[i].train_ids.push_back(r);
        for (size_t r = stop; r < len; ++r) folds[i].train_ids.push_back(r);
        for (size_t r = start; r < stop; ++r) folds[i].test_ids.push_back(r);
    }
    return folds;
}
std::vector<size_t> shuffle_permutation(size_t len, uint64_t seed) {
    std::vector<size_t> perm(len);
    for (size_t i = 0; i < len; ++i) perm[i] = i;
    if (len == 0) return perm;
    std::mt19937_64 rng(seed);
    for (size_t i = 0; i + 1 < len; ++i) {
        const size_t span = len - i - 1;  // j in [0, span): partner row i+1+j
        const size_t j = (size_t)(rng() % span);
        std::swap(perm[i], perm[i + 1 + j]);
    }
    return perm;
}

// ------------------------------------------------------------------------------------ Dataset
std::vector<std::pair<Dataset, Dataset>> Dataset::kfold(size_t k) const {
    std::vector<std::pair<Dataset, Dataset>> out;
    for (const Fold& f : kfold_ids(len(), k))
        out.emplace_back(Dataset(records_.select_rows(f.train_ids)), Dataset(records_.select_rows(f.test_ids)));
    return out;
}
std::vector<HostArray> Dataset::batch(size_t size, bool drop_last) const {
    std::vector<HostArray> out;
    for (const BatchRange& r : batch_ranges(len(), size, drop_last)) out.push_back(records_.slice_rows(r.start, r.rows));
    return out;
}
std::vector<Dataset> Dataset::split(const std::vector<size_t>& lengths) const {
    size_t total = 0;
    for (size_t l : lengths) total += l;
    if (total != len()) fail("error: input lengths do not cover the whole dataset.");  // lib.rs:95-97
    std::vector<Dataset> out;
    size_t start = 0;
    for (size_t l : lengths) {
        out.emplace_back(records_.slice_rows(start, l));
        start += l;
    }
    return out;
}
Dataset& Dataset::shuffle_with_seed(uint64_t seed) {
    records_ = records_.select_rows(shuffle_permutation(len(), seed));
    return *this;
}

LabeledDataset::LabeledDataset(HostArray records, HostArray labels) : records_(std::move(records)), labels_(std::move(labels)) {
    if (records_.rows() != labels_.rows()) fail("error: records and labels must have the same number of rows.");
}
std::vector<std::pair<LabeledDataset, LabeledDataset>> LabeledDataset::kfold(size_t k) const {
    std::vector<std::pair<LabeledDataset, LabeledDataset>> out;
    for (const Fold& f : kfold_ids(len(), k))
        out.emplace_back(LabeledDataset(records_.select_rows(f.train_ids), labels_.select_rows(f.train_ids)),
                         LabeledDataset(records_.select_rows(f.test_ids), labels_.select_rows(f.test_ids)));
    return out;
}
std::vector<std::pair<HostArray, HostArray>> LabeledDataset::batch(size_t size, bool drop_last) const {
    std::vector<std::pair<HostArray, HostArray>> out;
    for (const BatchRange& r : batch_ranges(len(), size, drop_last))
        out.emplace_back(records_.slice_rows(r.start, r.rows), labels_.slice_rows(r.start, r.rows));
    return out;
}
std::vector<LabeledDataset> LabeledDataset::split(const std::vector<size_t>& lengths) const {
    size_t total = 0;
    for (size_t l : lengths) total += l;
    if (total != len()) fail("error: input lengths do not cover the whole dataset.");  // lib.rs:584-586
    std::vector<LabeledDataset> out;
    size_t start = 0;
    for (size_t l : lengths) {
        out.emplace_back(records_.slice_rows(start, l), labels_.slice_rows(start, l));
        start += l;
    }
    return out;
}
LabeledDataset& LabeledDataset::shuffle_with_seed(uint64_t seed) {  // lib.rs:620-646: one permutation for both
    const auto perm = shuffle_permutation(len(), seed);
    records_ = records_.select_rows(perm);
    labels_ = labels_.select_rows(perm);
    return *this;
}

// ------------------------------------------------------------------------------------ CSV
namespace {
// csv-crate defaults that matter here: one record per line, `delimiter`, optional quotes, first line =
// headers unless disabled, every field parsed as f32.
std::vector<std::vector<float>> parse_csv(const std::string& text, bool headers, char delim) {
    std::vector<std::vector<float>> rows;
    std::istringstream in(text);
    std::string line;
    bool first = true;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (first && headers) { first = false; continue; }
        first = false;
        std::vector<float> row;
        size_t b = 0;
        while (b <= line.size()) {
            size_t e = line.find(delim, b);
            if (e == std::string::npos) e = line.size();
            size_t fb = b, fe = e;
            while (fb < fe && (line[fb] == ' ' || line[fb] == '"')) ++fb;
            while (fe > fb && (line[fe - 1] == ' ' || line[fe - 1] == '"')) --fe;
            float v = 0.f;
            auto r = std::from_chars(line.data() + fb, line.data() + fe, v);
            if (r.ec != std::errc() || r.ptr != line.data() + fe)
                fail("CSV deserialize error: field '" + line.substr(fb, fe - fb) + "' is not a number");
            row.push_back(v);
            b = e + 1;
        }
        rows.push_back(std::move(row));
    }
    return rows;
}
std::string read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail("No such file or directory: " + path);  // `File::open(src).unwrap()`
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
HostArray stack_rows(const std::vector<std::vector<float>>& rows, const Shape& record_shape, const char* what) {
    const size_t rl = numel_of(record_shape);
    if (rl == 0) fail("error: cannot handle empty records.");  // lib.rs:274-276
    HostArray out(stacked_shape(rows.size(), record_shape));
    for (size_t i = 0; i < rows.size(); ++i) {
        if (rows[i].size() != rl) fail(std::string("ShapeError: ") + what + " do not match the given shape");  // from_shape_vec().unwrap()
        std::memcpy(out.ptr() + i * rl, rows[i].data(), rl * sizeof(float));
    }
    return out;
}
}  // namespace

Dataset DataLoader::from_string(const std::string& text, const Shape& record_shape) const {
    return Dataset(stack_rows(parse_csv(text, headers_, delimiter_), record_shape, "records"));
}
Dataset DataLoader::from_csv(const std::string& path, const Shape& record_shape) const { return from_string(read_file(path), record_shape); }
LabeledDataLoader DataLoader::with_labels(const std::vector<size_t>& labels) const { return LabeledDataLoader(*this, labels); }

LabeledDataLoader::LabeledDataLoader(const DataLoader& base, std::vector<size_t> labels)
    : headers_(base.headers_), delimiter_(base.delimiter_), labels_(std::move(labels)) {
    if (labels_.empty()) fail("error: labels were not provided.");  // lib.rs:333-335
    std::sort(labels_.begin(), labels_.end());
    if (std::adjacent_find(labels_.begin(), labels_.end()) != labels_.end()) fail("error: duplicated labels.");  // :339-341
}
LabeledDataset LabeledDataLoader::from_string(const std::string& text, const Shape& record_shape, const Shape& label_shape) const {
    std::vector<std::vector<float>> rec, lab;
    for (const auto& row : parse_csv(text, headers_, delimiter_)) {
        std::vector<float> r, l;
        for (size_t c = 0; c < row.size(); ++c)  // lib.rs:316-321: label columns by index, the rest are inputs
            (std::binary_search(labels_.begin(), labels_.end(), c) ? l : r).push_back(row[c]);
        rec.push_back(std::move(r));
        lab.push_back(std::move(l));
    }
    return LabeledDataset(stack_rows(rec, record_shape, "records"), stack_rows(lab, label_shape, "labels"));
}
LabeledDataset LabeledDataLoader::from_csv(const std::string& path, const Shape& record_shape, const Shape& label_shape) const {
    return from_string(read_file(path), record_shape, label_shape);
}

// ------------------------------------------------------------------------------------ DeviceLoader
DeviceLoader::DeviceLoader(DevicePtr dev, const LabeledDataset& set, size_t batch_size, bool drop_last)
    : dev_(std::move(dev)), records_(set.records()), labels_(set.labels()), labeled_(true) {
    init(batch_size, drop_last);
}
DeviceLoader::DeviceLoader(DevicePtr dev, const Dataset& set, size_t batch_size, bool drop_last)
    : dev_(std::move(dev)), records_(set.records()) {
    init(batch_size, drop_last);
}
void DeviceLoader::init(size_t batch_size, bool drop_last) {
    ranges_ = batch_ranges(records_.rows(), batch_size, drop_last);
    if (ranges_.empty()) return;
    Shape xs = records_.shape(), ys = labels_.shape();
    xs[0] = (int)ranges_.front().rows;
    if (labeled_) ys[0] = xs[0];
    for (int s = 0; s < 2; ++s) {
        stage_x_[s] = std::make_shared<HipArray>(dev_, xs);
        if (labeled_) stage_y_[s] = std::make_shared<HipArray>(dev_, ys);
        check(nk_event_create(dev_->raw(), &ready_[s]));
        check(nk_event_create(dev_->raw(), &freed_[s]));
    }
    // the staging buffers come zeroed from the compute stream: the copy stream must not overtake that
    check(nk_event_record(freed_[0], 0));
    check(nk_event_record(freed_[1], 0));
    freed_valid_[0] = freed_valid_[1] = true;
    prefetch(0, 0);
}
DeviceLoader::~DeviceLoader() {
    if (dev_) dev_->sync();  // nothing may still read the pinned records / staging buffers
    for (int s = 0; s < 2; ++s) {
        if (ready_[s]) (void)nk_event_destroy(ready_[s]);
        if (freed_[s]) (void)nk_event_destroy(freed_[s]);
    }
}
void DeviceLoader::prefetch(size_t batch_no, int slot) {
    const BatchRange& r = ranges_[batch_no];
    if (freed_valid_[slot]) check(nk_stream_wait_event(dev_->raw(), 2, freed_[slot]));  // compute is done with the slot
    check(nk_upload_async(dev_->raw(), stage_x_[slot]->ptr(), records_.ptr() + r.start * records_.row_len(), r.rows * records_.row_len()));
    if (labeled_)
        check(nk_upload_async(dev_->raw(), stage_y_[slot]->ptr(), labels_.ptr() + r.start * labels_.row_len(), r.rows * labels_.row_len()));
    check(nk_event_record(ready_[slot], 2));
}
size_t DeviceLoader::next_into(const Var& x, const Var* y) {
    if (ranges_.empty()) return 0;
    if (cursor_ == ranges_.size()) {  // epoch finished: batch 0 of the next epoch is already on its way
        cursor_ = 0;
        return 0;
    }
    const int slot = (int)(cursor_ & 1);
    const BatchRange& r = ranges_[cursor_];
    const size_t nx = r.rows * records_.row_len();
    if (x.data->len() < nx) fail("DeviceLoader: destination smaller than a batch of records");
    if (labeled_ != (y != nullptr)) fail("DeviceLoader: labels destination does not match the dataset kind");
    check(nk_stream_wait_event(dev_->raw(), 0, ready_[slot]));
    check(nk_copy(dev_->raw(), x.data->ptr(), stage_x_[slot]->ptr(), nx));
    // a short last batch (drop_last = false) into a full-size leaf: the rows beyond it are ZEROED, never left holding the
    // previous batch's records (the reference's iterator yields a correctly sized view, neuronika-data/src/lib.rs:570;
    // a graph built for the full batch has to be rebuilt or masked by the caller for the ragged one)
    if (x.data->len() > nx) check(nk_fill(dev_->raw(), x.data->ptr() + nx, x.data->len() - nx, 0.f));
    if (labeled_) {
        const size_t ny = r.rows * labels_.row_len();
        if (y->data->len() < ny) fail("DeviceLoader: destination smaller than a batch of labels");
        check(nk_copy(dev_->raw(), y->data->ptr(), stage_y_[slot]->ptr(), ny));
        if (y->data->len() > ny) check(nk_fill(dev_->raw(), y->data->ptr() + ny, y->data->len() - ny, 0.f));
    }
    check(nk_event_record(freed_[slot], 0));
    freed_valid_[slot] = true;
    ++cursor_;
    // keep the copy stream one batch ahead (wrapping to the next epoch's first batch)
    const size_t nb = cursor_ == ranges_.size() ? 0 : cursor_;
    const int nslot = (int)(cursor_ == ranges_.size() ? 0 : (cursor_ & 1));
    // (with an odd number of batches the wrap re-uses the slot just consumed: its `freed` event, recorded
    // above, orders the overwrite after the copies out of it)
    prefetch(nb, nslot);
    return r.rows;
}
DeviceLoader::Batch DeviceLoader::next() {
    Batch b;
    if (ranges_.empty() || cursor_ == ranges_.size()) {
        next_into(Var(), nullptr);  // advances the epoch bookkeeping; touches nothing at the end of an epoch
        return b;
    }
    const BatchRange& r = ranges_[cursor_];
    Shape xs = records_.shape(), ys = labels_.shape();
    xs[0] = (int)r.rows;
    b.x = Var::leaf(std::make_shared<HipArray>(dev_, xs));
    if (labeled_) {
        ys[0] = (int)r.rows;
        b.y = Var::leaf(std::make_shared<HipArray>(dev_, ys));
    }
    b.rows = next_into(b.x, labeled_ ? &b.y : nullptr);
    return b;
}

}  // namespace data
}  // namespace neuronika
