//! Pinning kit, step 2 of 3: run the REFERENCE's own arithmetic over the inputs `tools/make_ref_inputs.py` wrote and
//! dump what it produces, so that `tests/test_reference_vectors.py` can compare the CPU oracle and the HIP path with
//! it (the only route off "parity unpinned").
//!
//!     cargo run --release --example dump_vectors -- DIR [ggml-model.bin]
//!
//! NOT COMPILED IN THIS REPOSITORY (no cargo here, crates not vendored: SURVEY.md 8c); the crate APIs below are written
//! from their published documentation [UPSTREAM-RECALL] at the versions the reference pins (src-tauri/Cargo.lock):
//! nnnoiseless 0.5.2 (:2825), rubato 0.16.2 (:4166), whisper-rs 0.16.0 (:6226).  Add them as dev-dependencies:
//!     [dev-dependencies] nnnoiseless = "=0.5.2"  rubato = "=0.16.2"  whisper-rs = "=0.16.0"  serde_json = "1"
use std::{env, fs, io::BufReader, path::Path};

fn read_f32(p: &Path) -> Vec<f32> {
    fs::read(p).expect("input file").chunks_exact(4).map(|b| f32::from_le_bytes([b[0], b[1], b[2], b[3]])).collect()
}
fn write_f32(p: &Path, v: &[f32]) {
    fs::write(p, v.iter().flat_map(|x| x.to_le_bytes()).collect::<Vec<u8>>()).expect("output file");
}

/// `process_frame` loop exactly as audio.rs:261-268 drives it: 480-sample frames, int16-range f32.
fn denoise(st: &mut nnnoiseless::DenoiseState, x: &[f32]) -> (Vec<f32>, Vec<f32>) {
    let (mut out, mut vad) = (vec![0f32; x.len()], Vec::new());
    for (i, o) in x.chunks_exact(nnnoiseless::FRAME_SIZE).zip(out.chunks_exact_mut(nnnoiseless::FRAME_SIZE)) {
        vad.push(st.process_frame(o, i));
    }
    (out, vad)
}

fn main() {
    let args: Vec<String> = env::args().collect();
    let dir = Path::new(args.get(1).expect("usage: dump_vectors DIR [ggml-model.bin]"));
    let man: serde_json::Value = serde_json::from_reader(BufReader::new(fs::File::open(dir.join("manifest.json")).unwrap())).unwrap();

    // ---- nnnoiseless::DenoiseState::process_frame (audio.rs:268) with models read from rnnoise-nu text files ----
    for case in man["rnnoise"].as_array().unwrap() {
        let model = nnnoiseless::RnnModel::from_read(BufReader::new(fs::File::open(dir.join(case["model"].as_str().unwrap())).unwrap()))
            .expect("rnnoise-nu model file");
        let mut st = nnnoiseless::DenoiseState::from_model(model);
        let (out, vad) = denoise(&mut st, &read_f32(&dir.join(case["in"].as_str().unwrap())));
        write_f32(&dir.join(case["ref_out"].as_str().unwrap()), &out);
        write_f32(&dir.join(case["ref_vad"].as_str().unwrap()), &vad);
    }
    {   // the model built into the crate: DenoiseState::new() (audio.rs:229)
        let b = &man["rnnoise_builtin"];
        let mut st = nnnoiseless::DenoiseState::new();
        let (out, vad) = denoise(&mut st, &read_f32(&dir.join(b["in"].as_str().unwrap())));
        write_f32(&dir.join(b["ref_out"].as_str().unwrap()), &out);
        write_f32(&dir.join(b["ref_vad"].as_str().unwrap()), &vad);
    }

    // ---- rubato::FftFixedIn(48000 -> 16000, chunk 1024, 1 sub-chunk, 1 channel), fed as commands/transcription.rs:314-357
    //      feeds it: 1024-sample calls, the last one zero-padded, every call's output appended ----
    {
        use rubato::Resampler;
        let r = &man["resampler"];
        let x = read_f32(&dir.join(r["in"].as_str().unwrap()));
        let mut rs = rubato::FftFixedIn::<f32>::new(48000, 16000, 1024, 1, 1).expect("resampler");
        let mut out = Vec::new();
        for chunk in x.chunks(1024) {
            let mut buf = chunk.to_vec();
            buf.resize(1024, 0.0);
            let y = rs.process(&[buf], None).expect("process");
            out.extend_from_slice(&y[0]);
        }
        write_f32(&dir.join(r["ref_out"].as_str().unwrap()), &out);
    }

    // ---- whisper.cpp through whisper-rs (what transcribe-rs' WhisperEngine wraps: managers/transcription.rs:138-141,
    //      183-185): greedy, language unset, timestamps on = TranscribeOptions::default() as far as it reaches whisper.cpp ----
    if let Some(model_path) = args.get(2) {
        use whisper_rs::{FullParams, SamplingStrategy, WhisperContext, WhisperContextParameters};
        let a = &man["asr"];
        let pcm = read_f32(&dir.join(a["in"].as_str().unwrap()));
        let ctx = WhisperContext::new_with_params(model_path, WhisperContextParameters::default()).expect("model");
        let mut state = ctx.create_state().expect("state");
        let mut p = FullParams::new(SamplingStrategy::Greedy { best_of: 1 });
        p.set_language(None);                 // auto-detect
        p.set_temperature_inc(0.0);           // no sampled fallback: deterministic (the library does not reproduce it)
        p.set_print_progress(false);
        p.set_print_realtime(false);
        state.full(p, &pcm).expect("whisper_full");
        let mut segs = Vec::new();
        let (mut tokens, mut text) = (Vec::new(), String::new());
        for i in 0..state.full_n_segments().unwrap() {
            let st = state.full_get_segment_text(i).unwrap();
            for j in 0..state.full_n_tokens(i).unwrap() { tokens.push(state.full_get_token_id(i, j).unwrap()); }
            segs.push(serde_json::json!({"t0": state.full_get_segment_t0(i).unwrap(), "t1": state.full_get_segment_t1(i).unwrap(), "text": st}));
            text.push_str(&st);
        }
        let lang = state.full_lang_id_from_state().unwrap_or(-1);
        fs::write(dir.join(a["ref"].as_str().unwrap()),
                  serde_json::to_vec_pretty(&serde_json::json!({"text": text, "segments": segs, "tokens": tokens, "lang_id": lang,
                                                                "t_unit": "centiseconds"})).unwrap()).unwrap();
    }
}
