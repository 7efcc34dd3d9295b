//! Acceptance test of the GPU back-end: every planned length must agree with the O(n^2) definition of the DFT
//! (`rustfft::algorithm::Dft`) under RustFFT's own tolerance — mean |difference| below 0.1 on inputs drawn from
//! [0, 10) — for f32 and f64, forward and inverse, through all three processing modes of the trait.
//! Needs a gfx950 device; without one `FftPlannerHip::new()` is `Err` and the tests report that and pass vacuously
//! (the same way RustFFT's ISA-specific tests behave on a machine without the ISA).

use rustfft::algorithm::Dft;
use rustfft::num_complex::Complex;
use rustfft::num_traits::Float;
use rustfft::{Fft, FftDirection, FftNum};
use rustfft_mi355::FftPlannerHip;

/// Small deterministic generator (no extra dev-dependencies): 64-bit LCG, upper bits -> [0, 10).
struct Lcg(u64);
impl Lcg {
    fn next_unit(&mut self) -> f64 {
        self.0 = self.0.wrapping_mul(6364136223846793005).wrapping_add(1442695040888963407);
        ((self.0 >> 11) as f64) / ((1u64 << 53) as f64)
    }
    fn signal<T: FftNum + Float>(&mut self, len: usize) -> Vec<Complex<T>> {
        (0..len).map(|_| Complex::new(T::from_f64(10.0 * self.next_unit()).unwrap(), T::from_f64(10.0 * self.next_unit()).unwrap())).collect()
    }
}

fn mean_abs_difference<T: FftNum + Float>(a: &[Complex<T>], b: &[Complex<T>]) -> f64 {
    assert_eq!(a.len(), b.len());
    if a.is_empty() {
        return 0.0;
    }
    let total: f64 = a.iter().zip(b).map(|(x, y)| (*x - *y).norm().to_f64().unwrap()).sum();
    total / a.len() as f64
}

fn check_len<T: FftNum + Float>(planner: &mut FftPlannerHip<T>, rng: &mut Lcg, len: usize, direction: FftDirection, batch: usize) {
    let fft = planner.plan_fft(len, direction);
    assert_eq!(fft.len(), len);
    assert_eq!(fft.fft_direction(), direction);
    let control = Dft::new(len, direction);
    let input: Vec<Complex<T>> = rng.signal(len * batch);
    let mut expected = input.clone();
    control.process(&mut expected);

    // in place
    let mut inplace = input.clone();
    let mut scratch = vec![Complex::new(T::zero(), T::zero()); fft.get_inplace_scratch_len()];
    fft.process_with_scratch(&mut inplace, &mut scratch);
    assert!(mean_abs_difference(&expected, &inplace) < 0.1, "in-place, len {} {:?}", len, direction);

    // out of place (input may be clobbered)
    let mut source = input.clone();
    let mut output = vec![Complex::new(T::zero(), T::zero()); input.len()];
    let mut scratch = vec![Complex::new(T::zero(), T::zero()); fft.get_outofplace_scratch_len()];
    fft.process_outofplace_with_scratch(&mut source, &mut output, &mut scratch);
    assert!(mean_abs_difference(&expected, &output) < 0.1, "out-of-place, len {} {:?}", len, direction);

    // immutable input
    let mut output = vec![Complex::new(T::zero(), T::zero()); input.len()];
    let mut scratch = vec![Complex::new(T::zero(), T::zero()); fft.get_immutable_scratch_len()];
    fft.process_immutable_with_scratch(&input, &mut output, &mut scratch);
    assert!(mean_abs_difference(&expected, &output) < 0.1, "immutable, len {} {:?}", len, direction);
}

fn sweep<T: FftNum + Float>() {
    let mut planner = match FftPlannerHip::<T>::new() {
        Ok(p) => p,
        Err(()) => {
            eprintln!("no gfx950 device: skipping");
            return;
        }
    };
    let mut rng = Lcg(0x1910_1143_1498_4148);
    for len in 1..1000 {
        for direction in [FftDirection::Forward, FftDirection::Inverse] {
            check_len::<T>(&mut planner, &mut rng, len, direction, 3);
        }
    }
    // beyond one workgroup: column-tile passes, prime tiles, multi-kernel Rader, Bluestein forms
    for len in [4096usize, 4099, 8192, 10403, 12289, 16384, 44100, 65536, 65537] {
        // the O(n^2) control is the cost here, not the GPU
        check_len::<T>(&mut planner, &mut rng, len, FftDirection::Forward, 1);
    }
}

#[test]
fn planned_lengths_match_the_definition_f32() {
    sweep::<f32>();
}

#[test]
fn planned_lengths_match_the_definition_f64() {
    sweep::<f64>();
}

#[test]
#[should_panic(expected = "multiple of FFT length")]
fn a_ragged_buffer_panics_like_rustfft() {
    let mut planner = match FftPlannerHip::<f32>::new() {
        Ok(p) => p,
        Err(()) => panic!("multiple of FFT length (no device: vacuous)"),
    };
    let fft = planner.plan_fft_forward(8);
    let mut buffer = vec![Complex::new(0f32, 0f32); 12];
    fft.process(&mut buffer);
}

#[test]
fn one_plan_serves_many_threads() {
    let mut planner = match FftPlannerHip::<f32>::new() {
        Ok(p) => p,
        Err(()) => return,
    };
    let fft = planner.plan_fft_forward(1 << 17); // a two-pass plan: the threads share its HBM workspace logic
    let control: Vec<Complex<f32>> = {
        let mut v = Lcg(7).signal::<f32>(1 << 17);
        fft.process(&mut v);
        v
    };
    let handles: Vec<_> = (0..4)
        .map(|_| {
            let fft = std::sync::Arc::clone(&fft);
            let control = control.clone();
            std::thread::spawn(move || {
                let mut v = Lcg(7).signal::<f32>(1 << 17);
                fft.process(&mut v);
                assert!(v == control, "results differ between threads");
            })
        })
        .collect();
    for h in handles {
        h.join().unwrap();
    }
}
