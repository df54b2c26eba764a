//! Safe layer over `ffi`: the process-wide device context, the factor handle that replaces
//! `covmat_cholesky: Cholesky<f64, Dynamic>` (src/gaussian_process/mod.rs:78) and the mapping of `fr_status` back to the
//! reference's panic texts.  Compiled only with `--features friedrich_mi355x`.
use super::ffi::*;
use crate::parameters::kernel::Kernel;
use nalgebra::{storage::Storage, Cholesky, DMatrix, DVector, Dynamic, Matrix, U1};
use std::ffi::CStr;
use std::os::raw::c_int;
use std::ptr;
use std::sync::OnceLock;

/// Below this many training rows every operation is faster on the host (measured cross-over).
pub const DEVICE_MIN_ROWS: usize = 128;

struct Ctx(*mut fr_ctx);
// The library takes the context's lock in every entry point (friedrich_amd.h, "Thread safety").
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {}

/// The device context, created on first use; `None` when no gfx950 GPU is visible (FR_NO_DEVICE): the caller keeps nalgebra.
pub fn context() -> Option<*mut fr_ctx>
{
    static CTX: OnceLock<Option<Ctx>> = OnceLock::new();
    CTX.get_or_init(|| {
           if unsafe { fr_abi_version() } != FR_ABI_VERSION
           {
               return None;
           }
           let mut ctx = ptr::null_mut();
           match unsafe { fr_ctx_create(&mut ctx, -1) }
           {
               FR_OK => Some(Ctx(ctx)),
               _ => None
           }
       })
       .as_ref()
       .map(|c| c.0)
}

pub fn last_error(ctx: *mut fr_ctx) -> String
{
    unsafe { CStr::from_ptr(fr_last_error(ctx)).to_string_lossy().into_owned() }
}

/// Any status the reference has no panic text for (HIP / RCCL / out of memory / shape): the library's own message.
pub fn check(ctx: *mut fr_ctx, status: c_int)
{
    if status != FR_OK
    {
        panic!("friedrich_amd (status {}): {}", status, last_error(ctx));
    }
}

/// Pointer + leading dimension of a column-major nalgebra matrix with any storage (`EMatrix::as_matrix()` has column stride =
/// capacity, src/algebra/extendable_matrix.rs:52-55).  Row stride must be 1, which holds for every matrix friedrich builds.
pub fn raw<R: nalgebra::Dim, C: nalgebra::Dim, S: Storage<f64, R, C>>(m: &Matrix<f64, R, C, S>) -> (*const f64, i64)
{
    let (row_stride, col_stride) = m.strides();
    assert_eq!(row_stride, 1, "friedrich_amd: matrices must be column-major with unit row stride");
    (m.as_ptr(), std::cmp::max(col_stride, m.nrows()) as i64)
}

/// Device program of a kernel, or `None` for a user-defined kernel (it stays on the nalgebra path).
pub fn program<K: Kernel>(kernel: &K) -> Option<fr_kprog>
{
    let mut prog = fr_kprog::default();
    if kernel.device_program(&mut prog) { Some(prog) } else { None }
}

pub fn push_leaf(prog: &mut fr_kprog, kind: i32, params: &[f64]) -> bool
{
    let at = prog.nops as usize;
    if at >= FR_KPROG_MAX_OPS || params.len() > 3
    {
        return false;
    }
    let mut op = fr_kernel_op { kind, nparams: params.len() as i32, params: [0.; 3] };
    op.params[..params.len()].copy_from_slice(params);
    prog.ops[at] = op;
    prog.nops += 1;
    true
}

pub fn push_op(prog: &mut fr_kprog, kind: i32) -> bool
{
    push_leaf(prog, kind, &[])
}

/// What `GaussianProcess::covmat_cholesky` becomes.
pub enum CholeskyHandle
{
    /// device-resident factor (+ the training inputs it was built from)
    Device(*mut fr_chol),
    /// user-defined kernel, no GPU, or a model below `DEVICE_MIN_ROWS`
    Host(Cholesky<f64, Dynamic>)
}

// One `fr_chol` may be used from several threads at once (GaussianProcess is Send + Sync: concurrent &self predicts).
unsafe impl Send for CholeskyHandle {}
unsafe impl Sync for CholeskyHandle {}

impl Drop for CholeskyHandle
{
    fn drop(&mut self)
    {
        if let CholeskyHandle::Device(h) = *self
        {
            unsafe { fr_chol_free(h) }
        }
    }
}

impl CholeskyHandle
{
    pub fn device(&self) -> Option<*mut fr_chol>
    {
        match self
        {
            CholeskyHandle::Device(h) => Some(*h),
            CholeskyHandle::Host(_) => None
        }
    }

    /// The factor as stock nalgebra would hold it (NaN above the diagonal, src/algebra/mod.rs:67): what serde writes, so
    /// that files stay loadable by friedrich 0.5.1 (mod.rs:58).
    pub fn to_host(&self) -> Cholesky<f64, Dynamic>
    {
        match self
        {
            CholeskyHandle::Host(c) => c.clone(),
            CholeskyHandle::Device(h) =>
            {
                let ctx = context().expect("friedrich_amd: device factor without a device");
                let (mut n, mut cap, mut d, mut ns, mut fc) = (0i64, 0i64, 0i64, 0i64, 0i64);
                check(ctx, unsafe { fr_chol_info(*h, &mut n, &mut cap, &mut d, &mut ns, &mut fc) });
                let mut l = DMatrix::<f64>::zeros(n as usize, n as usize);
                check(ctx, unsafe { fr_chol_download_l(*h, l.as_mut_ptr(), n, 1) });
                Cholesky::pack_dirty(l)
            }
        }
    }

    /// Deserialisation / a model that outgrew the host: the host factor moves to the device without being recomputed.
    pub fn to_device<S: Storage<f64, Dynamic, Dynamic>>(host: &Cholesky<f64, Dynamic>, inputs: &Matrix<f64, Dynamic, Dynamic, S>)
                                                        -> Option<CholeskyHandle>
    {
        let ctx = context()?;
        let l = host.l_dirty();
        let (lp, ldl) = raw(l);
        let (xp, ldx) = raw(inputs);
        let mut h = ptr::null_mut();
        let st = unsafe { fr_chol_upload_l(ctx, lp, l.nrows() as i64, ldl, xp, ldx, inputs.ncols() as i64, ldx, &mut h) };
        check(ctx, st);
        Some(CholeskyHandle::Device(h))
    }

    /// Number of pivots `cholesky_epsilon` replaced, and where ("pivot indices").
    pub fn substitutions(&self) -> Vec<usize>
    {
        match self
        {
            CholeskyHandle::Host(_) => vec![],
            CholeskyHandle::Device(h) =>
            {
                let ctx = context().expect("friedrich_amd: device factor without a device");
                let (mut n, mut cap, mut d, mut ns, mut fc) = (0i64, 0i64, 0i64, 0i64, 0i64);
                check(ctx, unsafe { fr_chol_info(*h, &mut n, &mut cap, &mut d, &mut ns, &mut fc) });
                let mut idx = vec![0i64; ns as usize];
                check(ctx, unsafe { fr_chol_substitutions(*h, idx.as_mut_ptr(), ns) });
                idx.into_iter().map(|i| i as usize).collect()
            }
        }
    }
}

#[cfg(feature = "friedrich_serde")]
impl serde::Serialize for CholeskyHandle
{
    fn serialize<Se: serde::Serializer>(&self, serializer: Se) -> Result<Se::Ok, Se::Error>
    {
        self.to_host().serialize(serializer)
    }
}

#[cfg(feature = "friedrich_serde")]
impl<'de> serde::Deserialize<'de> for CholeskyHandle
{
    /// Comes back as a host factor; `GaussianProcess`'s own `Deserialize` (which also has the training inputs) moves it over
    /// with `CholeskyHandle::to_device`.
    fn deserialize<De: serde::Deserializer<'de>>(deserializer: De) -> Result<Self, De::Error>
    {
        Cholesky::<f64, Dynamic>::deserialize(deserializer).map(CholeskyHandle::Host)
    }
}

/// `prior + vector` helpers of the call sites: a DVector the library writes into.
pub fn out_vector(len: usize) -> DVector<f64>
{
    DVector::<f64>::zeros(len)
}

/// Row views of the `Input` conversions are `1 x d`; the library wants `m x d` column-major: every `T::to_dmatrix` result is.
pub type RowView<'a, S> = Matrix<f64, U1, Dynamic, S>;
