// Host-side mirror of the `neuronika-data` crate (neuronika-data/src/lib.rs) plus the device half of
// its batch loop: `DeviceLoader` streams `dataset.batch(n)` into HBM through page-locked memory on a
// copy stream, double buffered, so the upload of batch k+1 overlaps the compute of batch k.
//
// Records live in host memory as dense row-major f32, axis 0 = sample (ndarray `Array<f32, D>`).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "neuronika.hpp"

namespace neuronika {
namespace data {

// Host tensor; page-locked when a HIP device is present (fast, truly asynchronous H2D), ordinary
// memory otherwise (CSV parsing / fold logic work without a GPU).
class HostArray {
   public:
    HostArray() = default;
    HostArray(Shape shape, const float* src);
    explicit HostArray(Shape shape);
    const Shape& shape() const { return shape_; }
    size_t len() const { return len_; }
    size_t rows() const { return shape_.empty() ? 0 : (size_t)shape_[0]; }
    size_t row_len() const { return rows() ? len_ / rows() : 0; }
    float* ptr() { return data_.get(); }
    const float* ptr() const { return data_.get(); }
    bool pinned() const { return pinned_; }
    std::vector<float> to_vec() const { return std::vector<float>(ptr(), ptr() + len_); }
    HostArray select_rows(const std::vector<size_t>& ids) const;  // `select(Axis(0), ids)`
    HostArray slice_rows(size_t start, size_t count) const;

   private:
    Shape shape_;
    size_t len_ = 0;
    bool pinned_ = false;
    std::shared_ptr<float> data_;
};

struct BatchRange {
    size_t start, rows;
};
// `axis_chunks_iter(Axis(0), size)` (+ `drop_last`, lib.rs:662-674: the last chunk is dropped only when
// its length differs from the first one)
std::vector<BatchRange> batch_ranges(size_t len, size_t size, bool drop_last);
// `SetKFold::compute_fold` lib.rs:692-719: step = 1 + (len-1)/k, test = [step*i, min(len, step*(i+1)))
struct Fold {
    std::vector<size_t> train_ids, test_ids;
};
std::vector<Fold> kfold_ids(size_t len, size_t k);
// Fisher-Yates in the reference's form (lib.rs:124-146): for i in 0..len-1 swap row i with row
// i+1+j, j uniform in [0, len-i-1).  (StdRng = ChaCha12 is not available here; the permutation is drawn
// from a 64-bit Mersenne twister seeded with `seed`, so it is reproducible but not the reference's.)
std::vector<size_t> shuffle_permutation(size_t len, uint64_t seed);

class Dataset {  // lib.rs:27-147
   public:
    explicit Dataset(HostArray records) : records_(std::move(records)) {}
    const HostArray& records() const { return records_; }
    size_t len() const { return records_.rows(); }
    bool is_empty() const { return len() == 0; }
    std::vector<std::pair<Dataset, Dataset>> kfold(size_t k) const;          // (train, test) per fold
    std::vector<HostArray> batch(size_t size, bool drop_last = false) const;  // copies; see DeviceLoader for the GPU path
    std::vector<Dataset> split(const std::vector<size_t>& lengths) const;
    Dataset& shuffle_with_seed(uint64_t seed);

   private:
    HostArray records_;
};

class LabeledDataset {  // lib.rs:508-647
   public:
    LabeledDataset(HostArray records, HostArray labels);
    const HostArray& records() const { return records_; }
    const HostArray& labels() const { return labels_; }
    size_t len() const { return records_.rows(); }
    bool is_empty() const { return len() == 0; }
    std::vector<std::pair<LabeledDataset, LabeledDataset>> kfold(size_t k) const;
    std::vector<std::pair<HostArray, HostArray>> batch(size_t size, bool drop_last = false) const;
    std::vector<LabeledDataset> split(const std::vector<size_t>& lengths) const;
    LabeledDataset& shuffle_with_seed(uint64_t seed);

   private:
    HostArray records_, labels_;
};

class LabeledDataLoader;
class DataLoader {  // lib.rs:149-299 (csv::ReaderBuilder: headers on, ',' delimiter by default)
   public:
    DataLoader& without_headers() { headers_ = false; return *this; }
    DataLoader& with_delimiter(char d) { delimiter_ = d; return *this; }
    LabeledDataLoader with_labels(const std::vector<size_t>& labels) const;
    Dataset from_csv(const std::string& path, const Shape& record_shape) const;
    Dataset from_string(const std::string& text, const Shape& record_shape) const;  // `from_reader`

   protected:
    friend class LabeledDataLoader;
    bool headers_ = true;
    char delimiter_ = ',';
};

class LabeledDataLoader {  // lib.rs:303-505
   public:
    LabeledDataLoader(const DataLoader& base, std::vector<size_t> labels);
    LabeledDataLoader& without_headers() { headers_ = false; return *this; }
    LabeledDataLoader& with_delimiter(char d) { delimiter_ = d; return *this; }
    LabeledDataset from_csv(const std::string& path, const Shape& record_shape, const Shape& label_shape) const;
    LabeledDataset from_string(const std::string& text, const Shape& record_shape, const Shape& label_shape) const;

   private:
    bool headers_;
    char delimiter_;
    std::vector<size_t> labels_;  // sorted column ids
};

// Device half of `for (x, y) in dataset.batch(n)`: batch k+1 is uploaded on the copy stream while the
// compute stream works on batch k.  `next_into` refills the leaves of a graph that was built once
// (their buffers are what the tape nodes hold); `next` returns fresh leaves for a per-batch graph.
class DeviceLoader {
   public:
    DeviceLoader(DevicePtr dev, const LabeledDataset& set, size_t batch_size, bool drop_last = true);
    DeviceLoader(DevicePtr dev, const Dataset& set, size_t batch_size, bool drop_last = true);
    ~DeviceLoader();
    DeviceLoader(const DeviceLoader&) = delete;
    DeviceLoader& operator=(const DeviceLoader&) = delete;
    size_t batches() const { return ranges_.size(); }
    // Copies the next batch into x (and y).  Returns the number of rows (0 = epoch finished; the next
    // call starts the following epoch).  A short last batch fills its rows and ZEROES the rest of the destination: the
    // caller must still use the returned row count (rebuild or mask the graph) - stale samples are never left behind.
    size_t next_into(const Var& x, const Var* y = nullptr);
    // Fresh leaves holding the next batch; `rows == 0` marks the end of the epoch.
    struct Batch {
        size_t rows = 0;
        Var x, y;
    };
    Batch next();

   private:
    void init(size_t batch_size, bool drop_last);
    void prefetch(size_t batch_no, int slot);
    DevicePtr dev_;
    HostArray records_, labels_;
    bool labeled_ = false;
    std::vector<BatchRange> ranges_;
    size_t cursor_ = 0;  // next batch to hand out
    Shared<HipArray> stage_x_[2], stage_y_[2];
    nk_event* ready_[2] = {nullptr, nullptr};  // upload of the slot finished (copy stream)
    nk_event* freed_[2] = {nullptr, nullptr};  // compute stream no longer reads the slot
    bool freed_valid_[2] = {false, false};
    int inflight_ = -1;  // batch number currently staged in slot (cursor_ & 1), -1 = none
};

}  // namespace data
}  // namespace neuronika
