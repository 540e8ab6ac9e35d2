//! MI355X (gfx950) backend: device arrays and nodes whose bodies are calls into `libneuronika_hip.so`.
//! Enabled by the `hip` feature (`#[cfg(feature = "hip")] pub mod hip;` in `lib.rs`, next to the `cuda` template).
mod device;
mod dp;
mod ffi;
mod hiparray;
mod hipvar;
mod node;
mod optimizer;

pub use {
    device::Device,
    dp::{Communicator, GradientSync, SyncEntry},
    hiparray::HipArray,
    hipvar::{manual_seed, HipVar, HipVarDiff, PaddingMode},
    optimizer::SGD,
};
