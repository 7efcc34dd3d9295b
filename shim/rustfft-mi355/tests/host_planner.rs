//! The host planner stays in charge: recipe tree, twiddle arithmetic and finished tables go through
//! `mi355fft_plan_create_ex`, and the results still match the definition.

use rustfft::algorithm::Dft;
use rustfft::num_complex::Complex;
use rustfft::{Fft, FftDirection};
use rustfft_mi355::{FftPlannerHip, HostPlannerOptions, RecipeTree};

fn host_twiddle(index: usize, fft_len: usize) -> Complex<f32> {
    // the arithmetic of rustfft's twiddle generation: angle in f64, rounded to T once
    let angle = -2.0 * std::f64::consts::PI * index as f64 / fft_len as f64;
    Complex::new(angle.cos() as f32, angle.sin() as f32)
}

#[test]
fn a_mixed_radix_recipe_sets_the_pass_heights() {
    let mut planner = match FftPlannerHip::<f32>::new() {
        Ok(p) => p,
        Err(()) => return,
    };
    // 2^20 as MixedRadix { left (width), right (height) }.  4096 x 256: no 4096-row column tile exists, so the library keeps
    // its own split and reports the family only (status 1); 512 x 2048 is realisable: a 2048-row pass, then a 512-row pass
    // (status 2)
    let leaf = |n: usize| Box::new(RecipeTree::Radix4 { k: 0, base: Box::new(RecipeTree::Dft(n)) });
    let unrealisable = RecipeTree::MixedRadix { left: leaf(1 << 12), right: leaf(1 << 8) };
    let realisable = RecipeTree::MixedRadix { left: leaf(1 << 9), right: leaf(1 << 11) };
    for (tree, want) in [(&unrealisable, 1), (&realisable, 2)] {
        let options = HostPlannerOptions { recipe: Some(tree), twiddle: Some(host_twiddle), ..Default::default() };
        let fft = planner.plan_fft_with(1 << 20, FftDirection::Forward, &options).expect("plan");
        assert_eq!(fft.recipe_status(), want, "{}", fft.describe());
        let mut v: Vec<Complex<f32>> = (0..1 << 20).map(|i| Complex::new((i % 7) as f32, (i % 5) as f32)).collect();
        let mut w = v.clone();
        fft.process(&mut v);
        planner.plan_fft_forward(1 << 20).process(&mut w);
        let err: f64 = v.iter().zip(&w).map(|(a, b)| (*a - *b).norm() as f64).sum::<f64>() / v.len() as f64;
        assert!(err < 0.1);
    }
}

#[test]
fn a_raders_recipe_runs_rader() {
    let mut planner = match FftPlannerHip::<f32>::new() {
        Ok(p) => p,
        Err(()) => return,
    };
    let tree = RecipeTree::Raders { inner: Box::new(RecipeTree::RadixN { factors: vec![7, 6], base: Box::new(RecipeTree::Butterfly(24)) }) };
    let options = HostPlannerOptions { recipe: Some(&tree), ..Default::default() };
    let fft = planner.plan_fft_with(1009, FftDirection::Inverse, &options).expect("plan");
    let mut v: Vec<Complex<f32>> = (0..1009).map(|i| Complex::new((i % 10) as f32, (i % 3) as f32)).collect();
    let mut w = v.clone();
    fft.process(&mut v);
    Dft::new(1009, FftDirection::Inverse).process(&mut w);
    let err: f64 = v.iter().zip(&w).map(|(a, b)| (*a - *b).norm() as f64).sum::<f64>() / v.len() as f64;
    assert!(err < 0.1);
}
