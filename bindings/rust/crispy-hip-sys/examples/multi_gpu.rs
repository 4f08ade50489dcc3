//! The stream-sharded multi-GPU split from Rust: one process, one `BatchDenoiser` per shard on
//! device `shard % device_count`, one host thread each, static block partition by stream id, no data-path collective.
//! The Rust twin of tests/c/multi_gpu.c (which the GPU test suite compiles and runs; there is no cargo in the build
//! image, so this file is checked by reading, like the rest of the crate).
//!
//!   cargo run --release --example multi_gpu -- model.txt in.f32 out.f32 <n_streams> <n_frames> <n_shards>
//!
//! in.f32 / out.f32: `[n_frames][n_streams][480]` raw f32 in int16 range.  Prints devices, shards and a checksum of the
//! output, which must not depend on the number of shards.
use std::path::Path;
use std::thread;

use crispy_hip_sys::{crispy_device_count, BatchDenoiser, CrispyError};

const FRAME: usize = 480;

fn main() -> Result<(), Box<dyn std::error::Error>> {
    let a: Vec<String> = std::env::args().collect();
    if a.len() != 7 {
        eprintln!("usage: {} model.txt in.f32 out.f32 n_streams n_frames n_shards", a[0]);
        std::process::exit(2);
    }
    let (b, t, r): (usize, usize, usize) = (a[4].parse()?, a[5].parse()?, a[6].parse()?);
    // SAFETY: no preconditions; never fails (0 without a gfx950 device).
    let n_dev = unsafe { crispy_device_count() }.max(0) as usize;
    if n_dev == 0 {
        return Err("no gfx950 device".into());
    }
    let raw = std::fs::read(&a[2])?;
    let input: Vec<f32> = raw.chunks_exact(4).map(|c| f32::from_le_bytes([c[0], c[1], c[2], c[3]])).collect();
    assert_eq!(input.len(), t * b * FRAME);
    let model = a[1].clone();
    // shard s owns streams [s b / r, (s + 1) b / r): crispy_amd/sharding.py `shard_range`
    let handles: Vec<_> = (0..r)
        .map(|s| {
            let (lo, hi) = (s * b / r, (s + 1) * b / r);
            let own = hi - lo;
            let mut x = vec![0f32; t * own * FRAME];
            for f in 0..t {
                x[f * own * FRAME..(f + 1) * own * FRAME].copy_from_slice(&input[(f * b + lo) * FRAME..(f * b + hi) * FRAME]);
            }
            let model = model.clone();
            thread::spawn(move || -> Result<(usize, usize, Vec<f32>), CrispyError> {
                let mut d = BatchDenoiser::from_model_file(Path::new(&model), own, (s % n_dev) as i32)?;
                let mut y = vec![0f32; x.len()];
                d.process(&x, &mut y, None, t)?;
                Ok((lo, hi, y))
            })
        })
        .collect();
    let mut out = vec![0f32; input.len()];
    for h in handles {
        let (lo, hi, y) = h.join().expect("shard thread panicked")?;
        let own = hi - lo;
        for f in 0..t {
            out[(f * b + lo) * FRAME..(f * b + hi) * FRAME].copy_from_slice(&y[f * own * FRAME..(f + 1) * own * FRAME]);
        }
    }
    let checksum: f64 = out.iter().map(|&v| v as f64).sum();
    std::fs::write(&a[3], out.iter().flat_map(|v| v.to_le_bytes()).collect::<Vec<u8>>())?;
    println!("{{\"devices\": {}, \"shards\": {}, \"streams\": {}, \"frames\": {}, \"checksum\": {:e}}}", n_dev, r, b, t, checksum);
    Ok(())
}
