//! BASELINE.md section 3, tier 1: the reference's own criterion harness shape (`bench-templates/src/lib.rs:29-138` times
//! `PCS::commit` and `PCS::open` separately on one dense polynomial, `hiding_bound = None`) for `MarlinKZG10<Bls12_381>` on
//! the CPU (ark-ec `msm_bigint`, rayon) and `HipMarlinKZG10<Bls12_381>` on the MI355X, at the degrees BASELINE.json names.
//! (`bench!` of bench-templates passes its `num_vars` to `setup` as the maximum degree, so degree 2^k needs this explicit form.)
//!
//!     PC_HIP_LIB_DIR=../../poly_commit_amd cargo bench --bench kzg_times            # needs a GPU
//!     KZG_BENCH_LOGS=12,16,20 cargo bench --bench kzg_times                          # subset of degrees
use ark_bls12_381::{Bls12_381, Fr};
use ark_pcs_bench_templates::test_sponge;
use ark_poly::{univariate::DensePolynomial, DenseUVPolynomial};
use ark_poly_commit::{marlin_pc::MarlinKZG10, LabeledPolynomial, PolynomialCommitment};
use ark_std::{test_rng, UniformRand};
use criterion::{criterion_group, criterion_main, BenchmarkId, Criterion};
use poly_commit_hip::HipMarlinKZG10;
use rand_chacha::{rand_core::SeedableRng, ChaCha20Rng};
use std::time::{Duration, Instant};

type Poly = DensePolynomial<Fr>;
type Cpu = MarlinKZG10<Bls12_381, Poly>;
type Hip = HipMarlinKZG10<Bls12_381, Poly>;

fn degrees() -> Vec<usize> {
    std::env::var("KZG_BENCH_LOGS").ok().map(|s| s.split(',').filter_map(|x| x.trim().parse().ok()).collect())
        .unwrap_or_else(|| vec![12, 16, 20, 24])       // configs[0] = 2^12 (CPU reference point), configs[1] = 2^20, north star = 2^24
}

fn bench_scheme<PCS>(c: &mut Criterion, name: &str)
where
    PCS: PolynomialCommitment<Fr, Poly, UniversalParams = <Cpu as PolynomialCommitment<Fr, Poly>>::UniversalParams>,
{
    let rng = &mut ChaCha20Rng::from_rng(test_rng()).unwrap();
    let max = *degrees().iter().max().unwrap();
    let pp = Cpu::setup(1 << max, None, rng).unwrap();          // one SRS, trimmed per degree (setup is the reference's)
    for lg in degrees() {
        let d = 1usize << lg;
        let (ck, _vk) = PCS::trim(&pp, d, 0, None).unwrap();
        let poly = LabeledPolynomial::new("p".to_string(), Poly::rand(d, rng), None, None);
        let point = Fr::rand(rng);
        // first use uploads the key and builds its window table (the `trim` work of the device path): outside the timing
        let (coms, states) = PCS::commit(&ck, [&poly], None).unwrap();
        let mut g = c.benchmark_group(format!("{} deg 2^{}", name, lg));
        g.sample_size(10).measurement_time(Duration::from_secs(if lg >= 22 { 20 } else { 5 }));
        g.bench_function(BenchmarkId::new("commit", lg), |b| {
            b.iter_custom(|iters| {
                let t = Instant::now();
                for _ in 0..iters {
                    let _ = PCS::commit(&ck, [&poly], None).unwrap();
                }
                t.elapsed()
            })
        });
        g.bench_function(BenchmarkId::new("open", lg), |b| {
            b.iter_custom(|iters| {
                let t = Instant::now();
                for _ in 0..iters {
                    let _ = PCS::open(&ck, [&poly], &coms, &point, &mut test_sponge::<Fr>(), &states, None).unwrap();
                }
                t.elapsed()
            })
        });
        g.finish();
    }
}

fn kzg(c: &mut Criterion) {
    bench_scheme::<Cpu>(c, "MarlinKZG10<Bls12_381> ark-ec CPU");
    bench_scheme::<Hip>(c, "HipMarlinKZG10<Bls12_381> MI355X");
}

criterion_group!(benches, kzg);
criterion_main!(benches);
