// tools/pin/dump_golden.rs -- DATA-GENERATING SCRIPT for a checkout of kchmck/p25rx (NOT built or run by this repository; see
// tools/pin/README.md).  Copy to <p25rx>/src/bin/dump_golden.rs and
//     cargo run --release --bin dump_golden -- pin_seed7.u8 baseband.f32le nid.jsonl
//
// Runs the reference's own stage sequence on a file of RTL-SDR bytes and writes what crosses its two seams:
//   baseband.f32le  the samples DemodTask::run sends as RecvEvent::Baseband (src/demod.rs:116): per 32 768-byte buffer
//                   (src/consts.rs:6) the LUT (src/demod.rs:82-84), decim_in_place (:87), bandpass.feed (:93), demod.feed
//                   (:109-111), avg.feed (:114) -- the loop body of src/demod.rs:70-117 with the channels removed
//   nid.jsonl       one line per MessageEvent::PacketNID that MessageReceiver::feed returns (src/recv.rs:207, 216-222) with the
//                   index of the baseband sample that produced it -- the loop of src/recv.rs:148-150 / src/replay.rs:44-57
// State persists across buffers exactly as in DemodTask (the objects live outside the loop, src/demod.rs:25-40).
extern crate demod_fm;
extern crate moving_avg;
extern crate num;
extern crate p25;
extern crate p25_filts;
extern crate rtlsdr_iq;
extern crate static_decimate;
extern crate static_fir;

use std::fs::File;
use std::io::{Read, Write};

use demod_fm::FmDemod;
use moving_avg::MovingAverage;
use num::complex::Complex32;
use num::traits::Zero;
use p25::message::receiver::{MessageEvent, MessageReceiver};
use p25_filts::{BandpassFir, DecimFir};
use rtlsdr_iq::IQ;
use static_decimate::Decimator;
use static_fir::FirFilter;

const BUF_BYTES: usize = 32768;                                                // src/consts.rs:6
const BUF_SAMPLES: usize = BUF_BYTES / 2;                                      // src/consts.rs:8

fn main() {
    let args: Vec<String> = std::env::args().collect();
    if args.len() != 4 {
        eprintln!("usage: dump_golden <in.u8> <baseband.f32le> <nid.jsonl>");
        std::process::exit(2);
    }
    let mut input = File::open(&args[1]).expect("unable to open input");
    let mut bbfile = File::create(&args[2]).expect("unable to create baseband file");
    let mut nidfile = File::create(&args[3]).expect("unable to create nid file");

    let mut decim: Decimator<DecimFir> = Decimator::new(5);                     // src/demod.rs:50
    let mut bandpass: FirFilter<BandpassFir> = FirFilter::new();               // src/demod.rs:51
    let mut avg: MovingAverage<f32> = MovingAverage::new(10);                   // src/demod.rs:52
    let mut demod = FmDemod::new(5000, 48000);                                  // src/demod.rs:54
    let mut msg = MessageReceiver::new();                                       // src/recv.rs:81

    let mut bytes = vec![0u8; BUF_BYTES];
    let mut samples = vec![Complex32::zero(); BUF_SAMPLES];
    let mut index: u64 = 0;
    loop {
        // a full buffer per iteration, like the dongle's callback (src/sdr.rs:25-33); a short tail is processed as it is
        let mut got = 0;
        while got < BUF_BYTES {
            let n = input.read(&mut bytes[got..]).expect("unable to read samples");
            if n == 0 { break; }
            got += n;
        }
        let pairs = got / 2;
        if pairs == 0 { break; }
        for i in 0..pairs {
            let s = (bytes[2 * i] as usize) | ((bytes[2 * i + 1] as usize) << 8);  // native-endian u16 on a little-endian host (src/demod.rs:74-76)
            samples[i] = IQ[s];                                                  // src/demod.rs:83
        }
        let len = decim.decim_in_place(&mut samples[..pairs]);                  // src/demod.rs:87
        let mut baseband = Vec::with_capacity(len);
        for i in 0..len {
            let y = bandpass.feed(samples[i]);                                   // src/demod.rs:93
            let f = demod.feed(y);                                               // src/demod.rs:110
            baseband.push(avg.feed(f));                                          // src/demod.rs:114
        }
        for &s in baseband.iter() {
            let raw: [u8; 4] = unsafe { std::mem::transmute::<f32, [u8; 4]>(s) };
            bbfile.write_all(&raw).expect("unable to write baseband");
            if let Some(event) = msg.feed(s) {                                   // src/recv.rs:207
                if let MessageEvent::PacketNID(nid) = event {                    // src/recv.rs:216
                    // {:?} of the fields the reference itself reads (src/policy.rs:96: nid.data_unit; the access code is the NAC)
                    writeln!(nidfile, "{{\"sample\":{},\"nid\":\"{:?}\"}}", index, nid).expect("unable to write nid");
                }
            }
            index += 1;
        }
        if got < BUF_BYTES { break; }
    }
}
