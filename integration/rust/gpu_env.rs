//! gpu_env.rs — `bourse_de::Env`-shaped safe wrapper over the raw FFI (`bourse_amd_sys.rs`).
//!
//! What a bourse maintainer adds to `crates/step_sim/src/` (next to `env.rs`) to run B books on an MI355X behind the
//! existing `Agent` trait.  Not compiled in this repository (no Rust toolchain in the build image); the C++17 mirror
//! `include/bourse_amd.hpp` has the same shape and IS compiled and tested (tests/cpp/env_mirror.cpp).
use super::bourse_amd_sys as sys;
use bourse_book::types::{Nanos, OrderId, Price, Side, Status, TraderId, Vol};
use bourse_book::OrderError;
use std::ffi::CStr;

fn last_error() -> String {
    unsafe { CStr::from_ptr(sys::bk_last_error()).to_string_lossy().into_owned() }
}

/// B independent `Env`s (or `n_books / assets` `MarketEnv<assets>`s) stepped in lockstep on one GPU.
pub struct GpuEnv {
    h: *mut sys::BkEnv,
    pub n_books: u32,
    pub levels: u32,
    tick_size: Price,
}

/// One book of a `GpuEnv` with the method set of `bourse_de::Env` (env.rs:58-295): hand `&mut GpuBook` to code written
/// against `Env` (e.g. `impl Agent for MyAgent { fn update(&mut self, env: &mut GpuBook, rng: &mut R) }`).
pub struct GpuBook<'a> {
    env: &'a mut GpuEnv,
    book: u32,
}

impl GpuEnv {
    /// `Env::<LEVELS>::new(start_time, tick_size, step_size, trading)` per book (env.rs:84-95); book b's shuffle RNG is
    /// `Xoroshiro128StarStar::seed_from_u64(seed + b)`.
    pub fn new(n_books: u32, seed: u64, start_time: Nanos, tick_size: Price, step_size: Nanos, trading: bool, levels: u32) -> Self {
        let cfg = sys::BkConfig {
            n_books, levels, start_time, tick_size, trading: trading as u32, step_size, seed, book_offset: 0,
            max_live_orders: 128, max_orders: 1 << 16, trade_capacity: 1 << 16, history_capacity: 1 << 10, device: 0, assets: 1,
        };
        let mut h = std::ptr::null_mut();
        let rc = unsafe { sys::bk_env_create(&cfg, &mut h) };
        assert_eq!(rc, sys::BK_OK, "bk_env_create: {}", last_error());
        Self { h, n_books, levels, tick_size }
    }
    pub fn book(&mut self, book: u32) -> GpuBook<'_> {
        assert!(book < self.n_books);
        GpuBook { env: self, book }
    }
    /// `Env::step` (env.rs:116-135) for every book: shuffle with the book's RNG, events at t0 + i, snapshot L2.
    pub fn step(&mut self) {
        let rc = unsafe { sys::bk_step(self.h) };
        assert_eq!(rc, sys::BK_OK, "bk_step: {}", last_error()); // the reference panics on an unknown order id too
    }
    /// `sim_runner(env, agents, seed, n_steps, _)` (runner.rs:46-69) with on-device `RandomAgents` groups.
    pub fn sim_runner(&mut self, agents: &[sys::BkRandomAgents], n_steps: u64) {
        unsafe {
            assert_eq!(sys::bk_set_random_agents(self.h, agents.len() as u32, agents.as_ptr()), sys::BK_OK);
            assert_eq!(sys::bk_run(self.h, n_steps), sys::BK_OK);
            assert_eq!(sys::bk_env_sync(self.h), sys::BK_OK);
        }
    }
    /// `Env::level_2_data` of books [first, first + n) in the numpy layout (rust/src/step_sim_numpy.rs:351-368).
    pub fn level_2_data(&mut self, first: u32, n: u32) -> Vec<u32> {
        let w = unsafe { sys::bk_l2_width(self.h) } as usize;
        let mut out = vec![0u32; w * n as usize];
        assert_eq!(unsafe { sys::bk_level2(self.h, first, n, out.as_mut_ptr()) }, sys::BK_OK);
        out
    }
}

impl<'a> GpuBook<'a> {
    /// `Env::place_order` (env.rs:166-176): `Err(PriceError)` creates and queues nothing.
    pub fn place_order(&mut self, side: Side, vol: Vol, trader_id: TraderId, price: Option<Price>) -> Result<OrderId, OrderError> {
        let mut id = 0u64;
        let rc = unsafe {
            sys::bk_place_order(self.env.h, self.book, bool::from(side) as i32, vol, trader_id, price.is_some() as i32,
                                price.unwrap_or(0), &mut id)
        };
        match rc {
            sys::BK_OK => Ok(id as OrderId),
            sys::BK_PRICE_NOT_TICK_MULTIPLE => Err(OrderError::PriceError { price: price.unwrap(), tick_size: self.env.tick_size }),
            _ => panic!("bk_place_order: {}", last_error()),
        }
    }
    pub fn cancel_order(&mut self, order_id: OrderId) {
        unsafe { sys::bk_cancel_order(self.env.h, self.book, order_id as u64) }; // env.rs:189-191
    }
    pub fn modify_order(&mut self, order_id: OrderId, new_price: Option<Price>, new_vol: Option<Vol>) {
        unsafe {
            sys::bk_modify_order(self.env.h, self.book, order_id as u64, new_price.is_some() as i32, new_price.unwrap_or(0),
                                 new_vol.is_some() as i32, new_vol.unwrap_or(0))
        }; // env.rs:208-219
    }
    pub fn order_status(&mut self, order_id: OrderId) -> Status {
        let mut s = 0u8;
        assert_eq!(unsafe { sys::bk_order_status(self.env.h, self.book, order_id as u64, &mut s) }, sys::BK_OK);
        match s { 0 => Status::New, 1 => Status::Active, 2 => Status::Filled, 3 => Status::Cancelled, _ => Status::Rejected }
    }
}

impl Drop for GpuEnv {
    fn drop(&mut self) {
        unsafe { sys::bk_env_destroy(self.h) }
    }
}
