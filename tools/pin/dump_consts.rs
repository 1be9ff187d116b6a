// tools/pin/dump_consts.rs -- DATA-GENERATING SCRIPT for a checkout of kchmck/p25rx (NOT built or run by this repository:
// the image has no Rust toolchain; see tools/pin/README.md).  Copy to <p25rx>/src/bin/dump_consts.rs and
//     cargo run --release --bin dump_consts > consts.json
//
// Every number docs/SPEC.md marks "build-defined" lives in crates the reference pins in its Cargo.lock
// (Cargo.lock:134-136, 320-322, 442-444, 458-460, 575-577, 655-657, 664-666).  This program dumps them BEHAVIOURALLY -- it
// drives exactly the objects DemodTask constructs (src/demod.rs:49-54) through exactly the calls DemodTask::run makes
// (src/demod.rs:83, 87, 93, 110, 114) with impulses and probe samples, so it does not depend on any accessor of those crates
// that the reference itself does not use.  tools/pin/load_pin.py turns the responses back into tables and constants.
//
// Output: ONE JSON object
//   "iq_lut"        65536 x [re, im]: rtlsdr_iq::IQ[s] for every u16 (src/demod.rs:83)
//   "decim_impulse" for p = 0..9: the outputs of Decimator<DecimFir>::decim_in_place (src/demod.rs:87) on a fresh
//                   decimator for a 640-sample buffer that is 1 + 0j at index p and 0 elsewhere (real parts, `len` outputs)
//   "chan_impulse"  256 outputs of a fresh FirFilter<BandpassFir>::feed (src/demod.rs:93) for 1 + 0j followed by zeros
//   "avg_impulse"   32 outputs of a fresh MovingAverage::new(10).feed (src/demod.rs:114) for 1.0 followed by zeros
//   "fm_probe"      FmDemod::new(5000, 48000).feed (src/demod.rs:54, 110) on a fresh demodulator: for k = 0..359 the output
//                   for the second sample of the pair (1 + 0j, e^{j 2 pi k / 360}) -> [k, output]
//   "fm_pairs"      a few hundred pseudo-random pairs (prev, cur) -> [prev.re, prev.im, cur.re, cur.im, output] (bit patterns
//                   as u32 too): what tools/pin/load_pin.py compares the oracle's discriminator with, value by value
extern crate demod_fm;
extern crate moving_avg;
extern crate num;
extern crate p25_filts;
extern crate rtlsdr_iq;
extern crate static_decimate;
extern crate static_fir;

use demod_fm::FmDemod;
use moving_avg::MovingAverage;
use num::complex::Complex32;
use num::traits::Zero;
use p25_filts::{BandpassFir, DecimFir};
use rtlsdr_iq::IQ;
use static_decimate::Decimator;
use static_fir::FirFilter;

fn bits(x: f32) -> u32 {
    unsafe { std::mem::transmute::<f32, u32>(x) }
}

fn main() {
    let mut out = String::from("{");

    out.push_str("\"iq_lut\":[");
    for s in 0..65536usize {
        let c = IQ[s];
        if s > 0 { out.push(','); }
        out.push_str(&format!("[{},{}]", bits(c.re), bits(c.im)));
    }
    out.push_str("],");

    out.push_str("\"decim_impulse\":[");
    for p in 0..10usize {
        let mut decim: Decimator<DecimFir> = Decimator::new(5);                 // src/demod.rs:50
        let mut buf = vec![Complex32::zero(); 640];
        buf[p] = Complex32::new(1.0, 0.0);
        let len = decim.decim_in_place(&mut buf[..]);                           // src/demod.rs:87
        if p > 0 { out.push(','); }
        out.push('[');
        for m in 0..len {
            if m > 0 { out.push(','); }
            out.push_str(&format!("{}", bits(buf[m].re)));
        }
        out.push(']');
    }
    out.push_str("],");

    out.push_str("\"chan_impulse\":[");
    let mut bandpass: FirFilter<BandpassFir> = FirFilter::new();               // src/demod.rs:51
    for n in 0..256usize {
        let x = if n == 0 { Complex32::new(1.0, 0.0) } else { Complex32::zero() };
        let y = bandpass.feed(x);                                               // src/demod.rs:93
        if n > 0 { out.push(','); }
        out.push_str(&format!("{}", bits(y.re)));
    }
    out.push_str("],");

    out.push_str("\"avg_impulse\":[");
    let mut avg: MovingAverage<f32> = MovingAverage::new(10);                   // src/demod.rs:52
    for n in 0..32usize {
        let y = avg.feed(if n == 0 { 1.0 } else { 0.0 });                       // src/demod.rs:114
        if n > 0 { out.push(','); }
        out.push_str(&format!("{}", bits(y)));
    }
    out.push_str("],");

    out.push_str("\"fm_probe\":[");
    for k in 0..360usize {
        let mut fm = FmDemod::new(5000, 48000);                                 // src/demod.rs:54 (BASEBAND_SAMPLE_RATE, src/consts.rs:13)
        let a = 2.0 * std::f64::consts::PI * (k as f64) / 360.0;
        fm.feed(Complex32::new(1.0, 0.0));
        let y = fm.feed(Complex32::new(a.cos() as f32, a.sin() as f32));        // src/demod.rs:110
        if k > 0 { out.push(','); }
        out.push_str(&format!("[{},{}]", k, bits(y)));
    }
    out.push_str("],");

    out.push_str("\"fm_pairs\":[");
    let mut st: u32 = 0x2545F491;
    let mut next = || { st ^= st << 13; st ^= st >> 17; st ^= st << 5; (st as f64 / 4294967296.0 - 0.5) as f32 };
    for k in 0..512usize {
        let prev = Complex32::new(next(), next());
        let cur = Complex32::new(next(), next());
        let mut fm = FmDemod::new(5000, 48000);
        fm.feed(prev);
        let y = fm.feed(cur);
        if k > 0 { out.push(','); }
        out.push_str(&format!("[{},{},{},{},{}]", bits(prev.re), bits(prev.im), bits(cur.re), bits(cur.im), bits(y)));
    }
    out.push_str("]}");
    println!("{}", out);
}
