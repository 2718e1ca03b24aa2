#!/usr/bin/env node
/* make_golden.js — cuts the golden fixtures in this directory by running the REFERENCE
 * JavaScript (read from /root/reference, never copied) under Node in the build container.
 *
 *   node tests/golden/make_golden.js small            -> golden_small.json, kat.json
 *   node tests/golden/make_golden.js big <jobname>    -> golden_big_<jobname>.json
 *   node tests/golden/make_golden.js api              -> golden_api.json (error texts, size-less stream input, mixed-level multistream)
 *
 * The reference is loaded with vm.runInContext (the joined files define top-level vars and
 * export nothing: J/Bzip2_joined_.js:3-10).  Oracle = reference under Node >= 11 (stable
 * Array.prototype.sort, SURVEY.md Q16); process.version is recorded in every fixture file.
 * Synthetic inputs come from tools/textgen.c (binary built to /tmp/textgen by the caller).
 *
 * Input recipes (mirrored by tests/recipes.py):
 *   {kind:'file', name}                      reference test fixture, copied to tests/golden/data/
 *   {kind:'textgen', n, seed}                tools/textgen.c stream
 *   {kind:'repeat', unit_hex, n}             unit repeated/truncated to n bytes
 *   {kind:'xorshift', n, seed, mask, add}    bytes (xorshift32 >>> 24) & mask) + add
 *   {kind:'range256', n}                     i & 255
 *   {kind:'concat', parts:[recipe...]}
 */
'use strict';
const fs = require('fs');
const vm = require('vm');
const path = require('path');
const crypto = require('crypto');
const { execFileSync } = require('child_process');

const REF = '/root/reference';
const NPM_TEST = REF + '/complete original reference from npm/node_modules/compressjs/test';
const HERE = __dirname;
const TEXTGEN = process.env.TEXTGEN || '/tmp/textgen';

function loadJoined(file, names) {
  const src = fs.readFileSync(path.join(REF, file), 'utf8');
  // console.assert is bound as ASSERT inside the SA-IS loops; a no-op assert changes nothing
  // in the output and makes the oracle ~2x faster.
  const fakeConsole = { assert: function () {}, log: console.log, error: console.error, warn: console.warn };
  const ctx = vm.createContext({ console: fakeConsole });
  vm.runInContext(src + ';\n' + names.map(n => 'this.__' + n + '=' + n + ';').join(''), ctx, { filename: file });
  const out = {};
  names.forEach(n => { out[n] = ctx['__' + n]; });
  return out;
}

function sha256(buf) { return crypto.createHash('sha256').update(buf).digest('hex'); }

function build(recipe) {
  switch (recipe.kind) {
    case 'file': return new Uint8Array(fs.readFileSync(path.join(NPM_TEST, recipe.name)));
    case 'textgen': {
      const tmp = '/tmp/textgen_' + recipe.n + '_' + recipe.seed + '.bin';
      if (!fs.existsSync(tmp) || fs.statSync(tmp).size !== recipe.n) {
        const fd = fs.openSync(tmp, 'w');
        execFileSync(TEXTGEN, [String(recipe.n), String(recipe.seed)], { stdio: ['ignore', fd, 'inherit'] });
        fs.closeSync(fd);
      }
      return new Uint8Array(fs.readFileSync(tmp));
    }
    case 'repeat': {
      const unit = Buffer.from(recipe.unit_hex, 'hex');
      const out = new Uint8Array(recipe.n);
      for (let i = 0; i < recipe.n; i++) out[i] = unit[i % unit.length];
      return out;
    }
    case 'xorshift': {
      let s = recipe.seed >>> 0;
      const out = new Uint8Array(recipe.n);
      const mask = recipe.mask === undefined ? 255 : recipe.mask, add = recipe.add || 0;
      for (let i = 0; i < recipe.n; i++) {
        s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0;
        out[i] = (((s >>> 24) & mask) + add) & 255;
      }
      return out;
    }
    case 'range256': {
      const out = new Uint8Array(recipe.n);
      for (let i = 0; i < recipe.n; i++) out[i] = i & 255;
      return out;
    }
    case 'concat': {
      const parts = recipe.parts.map(build);
      const n = parts.reduce((a, p) => a + p.length, 0);
      const out = new Uint8Array(n);
      let o = 0;
      parts.forEach(p => { out.set(p, o); o += p.length; });
      return out;
    }
  }
  throw new Error('bad recipe ' + JSON.stringify(recipe));
}

function runCase(mods, name, recipe, algo, level, keepHexBelow) {
  const input = build(recipe);
  const t0 = Date.now();
  const out = mods[algo].compressFile(input, null, level);
  const dt = (Date.now() - t0) / 1000;
  const rec = { name, recipe, algo, level, in_len: input.length, in_sha256: sha256(input),
                out_len: out.length, out_sha256: sha256(out), ref_seconds: dt };
  if (out.length <= (keepHexBelow || 0)) rec.out_hex = Buffer.from(out).toString('hex');
  // decoder cross-check with the reference's own decoder
  if (input.length <= 3000000) {
    const back = mods[algo].decompressFile(out);
    if (back.length !== input.length || sha256(back) !== rec.in_sha256) throw new Error('reference round trip failed: ' + name);
  }
  process.stderr.write(name + ' ' + algo + ' -' + level + ': ' + input.length + ' -> ' + out.length + ' (' + dt.toFixed(2) + ' s)\n');
  return rec;
}

const BZ_MODS = ['Bzip2', 'BWT', 'HuffmanAllocator', 'CRC32', 'Util'];
const BW_MODS = ['BWTC', 'RangeCoder', 'FenwickModel', 'DefSumModel', 'LogDistanceModel', 'NoModel', 'BWT'];

function smallCases() {
  const cases = [];
  const T = (n, seed) => ({ kind: 'textgen', n, seed });
  const rep = (hex, n) => ({ kind: 'repeat', unit_hex: hex, n });
  const rnd = (n, seed, mask, add) => ({ kind: 'xorshift', n, seed, mask, add });
  const lit = s => ({ kind: 'repeat', unit_hex: Buffer.from(s, 'ascii').toString('hex') || '00', n: s.length });
  // tiny strings (full streams kept)
  ['', 'a', 'aaaa', 'abab', 'banana', 'aaaaa', 'abracadabra', 'mississippi'].forEach((s, i) =>
    cases.push(['tiny_' + (s || 'empty'), s === '' ? rep('00', 0) : lit(s), [1, 9]]));
  // reference fixtures
  for (let i = 0; i <= 5; i++) cases.push(['sample' + i, { kind: 'file', name: 'sample' + i + '.ref' }, i === 5 ? [1, 9] : [1, 5, 9]]);
  // adversarial (SURVEY.md §8(d))
  cases.push(['zeros_300000', rep('00', 300000), [1, 9]]);
  cases.push(['ab_10000', rep('6162', 10000), [1, 9]]);
  cases.push(['ab_250001', rep('6162', 250001), [1, 9]]);
  cases.push(['abc_period3_100000', rep('616263', 100000), [1]]);
  cases.push(['range256_70000', { kind: 'range256', n: 70000 }, [1, 9]]);
  cases.push(['random_250000', rnd(250000, 12345, 255, 0), [1, 2, 9]]);
  cases.push(['random4sym_200000', rnd(200000, 777, 3, 97), [1, 9]]);
  cases.push(['random2sym_150000', rnd(150000, 4242, 1, 48), [1, 9]]);
  // Q2: 4th byte of a run lands in the last slot of a level-1 block (99,981)
  cases.push(['q2_run_at_block_end', { kind: 'concat', parts: [rnd(99977, 99, 63, 32), rep('00', 4), rep('41', 50), rnd(1000, 5, 63, 32)] }, [1]]);
  // count byte fills the block
  cases.push(['q2_count_fills_block', { kind: 'concat', parts: [rnd(99976, 98, 63, 32), rep('00', 40), rnd(1000, 6, 63, 32)] }, [1]]);
  // run crossing a block end earlier
  cases.push(['q2_run_crosses_block', { kind: 'concat', parts: [rnd(99979, 97, 63, 32), rep('7a', 700), rnd(500, 7, 63, 32)] }, [1]]);
  // exact block multiples (no runs: mask 63 + 32 can repeat by chance, so use a counter-ish stream)
  cases.push(['exact_block_99981', { kind: 'concat', parts: [rep('6162636465666768696a', 99981)] }, [1]]);
  cases.push(['exact_2blocks_199962', { kind: 'concat', parts: [rep('6162636465666768696a6b', 199962)] }, [1]]);
  cases.push(['long_runs_mixed', { kind: 'concat', parts: [rep('61', 255), rep('62', 256), rep('63', 259), rep('64', 4), rep('65', 5), rep('66', 1000), rep('67', 3), rep('61', 260), rep('6162', 9), rep('00', 511)] }, [1, 9]]);
  cases.push(['run4_at_eof', { kind: 'concat', parts: [lit('xyz'), rep('71', 4)] }, [9]]);
  // synthetic text, multi-block at -1 and single block at -9
  cases.push(['textgen_1000000_s1', T(1000000, 1), [1, 5, 9]]);
  cases.push(['textgen_3000000_s2', T(3000000, 2), [1, 9]]);
  cases.push(['textgen_65536_s1', T(65536, 1), [1, 9]]);
  return cases;
}

function kats(bz, bw) {
  // known answers for individual stages, straight from the reference modules
  const out = { node: process.version, bwt_cyclic: [], bwt_sentinel: [], huffman_alloc: [], crc: [], rle2_note: 'see full streams' };
  const strs = ['bcababa', 'ABCDEFGHIJKLMNOPQRSTUVWXYZ', 'banana', 'abab', 'aaaa', 'abababababab', 'mississippi', 'a', 'ab', 'ba',
                'SIX.MIXED.PIXIES.SIFT.SIXTY.PIXIE.DUST.BOXES', 'abcabcabcabd', 'zzzzzzzzzy'];
  strs.forEach(s => {
    const T = new Uint8Array(Buffer.from(s, 'ascii'));
    const U = new Uint8Array(T.length);
    const p = bz.BWT.bwtransform2(T, U, T.length, 256);
    out.bwt_cyclic.push({ input_hex: Buffer.from(T).toString('hex'), out_hex: Buffer.from(U).toString('hex'), pidx: p });
    const U2 = new Uint8Array(T.length), A = new Int32Array(T.length);
    const p2 = bz.BWT.bwtransform(T, U2, A, T.length, 256);
    out.bwt_sentinel.push({ input_hex: Buffer.from(T).toString('hex'), out_hex: Buffer.from(U2).toString('hex'), pidx: p2 });
  });
  // (ab)^5000 -> 4999 (Q4)
  {
    const T = build({ kind: 'repeat', unit_hex: '6162', n: 10000 });
    const U = new Uint8Array(T.length);
    out.bwt_cyclic.push({ recipe: { kind: 'repeat', unit_hex: '6162', n: 10000 }, out_sha256: null, pidx: bz.BWT.bwtransform2(T, U, T.length, 256) });
    out.bwt_cyclic[out.bwt_cyclic.length - 1].out_sha256 = sha256(U);
  }
  const freqs = [[0, 0, 0, 0, 0], [0, 0, 0], [0, 1], [0, 0, 0, 1, 1000], [0, 0, 0, 0, 0, 0, 0, 0, 1], [1, 1, 1, 1, 1], [1, 1], [1],
                 [0, 0, 1, 1, 1, 1], [1, 2, 3, 4, 5, 6, 7, 8, 9, 10], [5, 5, 5, 5, 5, 5, 5, 5]];
  const pow2 = []; for (let i = 0; i <= 22; i++) pow2.push(1 << i);
  freqs.push(pow2);
  const fib = [0, 1]; while (fib.length < 37) fib.push(fib[fib.length - 1] + fib[fib.length - 2]);
  [[36, 20], [22, 20], [21, 20], [36, 6]].forEach(([k, lim]) => freqs.push({ f: fib.slice(0, k), lim }));
  // pseudo-random sorted frequency vectors with many zeros (Q10)
  let s = 2463534242;
  const r = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s; };
  for (let t = 0; t < 24; t++) {
    const n = 3 + (r() % 256);
    const f = [];
    for (let i = 0; i < n; i++) { const z = r() % 4; f.push(z === 0 ? 0 : (r() % (1 << (1 + (r() % 20))))); }
    f.sort((a, b) => a - b);
    freqs.push(f);
  }
  freqs.forEach(f => {
    const lim = f.lim || 20, arr = (f.f || f).slice();
    const inp = arr.slice();
    bz.HuffmanAllocator.allocateHuffmanCodeLengths(arr, lim);
    out.huffman_alloc.push({ freq_sorted: inp, limit: lim, lengths: arr });
  });
  ['', 'a', '123456789', 'The quick brown fox'].forEach(sx => {
    const c = new bz.CRC32();
    for (let i = 0; i < sx.length; i++) c.updateCRC(sx.charCodeAt(i));
    out.crc.push({ ascii: sx, crc: c.getCRC() });
  });
  return out;
}

function copyFixtures() {
  const dst = path.join(HERE, 'data');
  fs.mkdirSync(dst, { recursive: true });
  fs.readdirSync(NPM_TEST).forEach(f => {
    if (/^sample\d\.(ref|bz2|bzt|\d+)$/.test(f)) fs.copyFileSync(path.join(NPM_TEST, f), path.join(dst, f));
  });
}

function main() {
  const mode = process.argv[2];
  const bz = loadJoined('Bzip2_joined_.js', BZ_MODS);
  const bw = loadJoined('BWTC_joined_.js', BW_MODS);
  const mods = { Bzip2: bz.Bzip2, BWTC: bw.BWTC };
  if (mode === 'small') {
    copyFixtures();
    const recs = [];
    smallCases().forEach(([name, recipe, levels]) => {
      levels.forEach(level => {
        ['Bzip2', 'BWTC'].forEach(algo => recs.push(runCase(mods, name, recipe, algo, level, 400)));
      });
    });
    fs.writeFileSync(path.join(HERE, 'golden_small.json'), JSON.stringify({ node: process.version, generator_version: 1, cases: recs }, null, 1));
    fs.writeFileSync(path.join(HERE, 'kat.json'), JSON.stringify(kats(bz, bw), null, 1));
  } else if (mode === 'big') {
    const jobs = {
      bzip2_9_100m: ['textgen_100000000_s1', { kind: 'textgen', n: 100000000, seed: 1 }, 'Bzip2', 9],
      bzip2_1_100m: ['textgen_100000000_s1', { kind: 'textgen', n: 100000000, seed: 1 }, 'Bzip2', 1],
      bwtc_9_100m: ['textgen_100000000_s1', { kind: 'textgen', n: 100000000, seed: 1 }, 'BWTC', 9],
      bzip2_9_10m: ['textgen_10000000_s1', { kind: 'textgen', n: 10000000, seed: 1 }, 'Bzip2', 9],
      bzip2_1_10m: ['textgen_10000000_s1', { kind: 'textgen', n: 10000000, seed: 1 }, 'Bzip2', 1],
      bwtc_9_10m: ['textgen_10000000_s1', { kind: 'textgen', n: 10000000, seed: 1 }, 'BWTC', 9],
      bwtc_9_1g: ['textgen_1073741824_s1', { kind: 'textgen', n: 1073741824, seed: 1 }, 'BWTC', 9],
      bzip2_9_1g: ['textgen_1073741824_s1', { kind: 'textgen', n: 1073741824, seed: 1 }, 'Bzip2', 9],
    };
    const job = jobs[process.argv[3]];
    if (!job) throw new Error('unknown job; one of ' + Object.keys(jobs).join(' '));
    const rec = runCase(mods, job[0], job[1], job[2], job[3], 0);
    fs.writeFileSync(path.join(HERE, 'golden_big_' + process.argv[3] + '.json'), JSON.stringify({ node: process.version, generator_version: 1, cases: [rec] }, null, 1));
  } else if (mode === 'api') {
    // Boundary behaviour of the reference recorded as data: what it throws for damaged .bz2 input (message, errorCode,
    // error class), what BWTC writes for a stream input without .size (SURVEY W1) and that Bunzip.decode restarts with a
    // new level inside a multistream file (J/Bzip2_joined_.js:1787-1792).
    const hex = b => Buffer.from(b).toString('hex');
    const banana = Buffer.from('banana');
    const good = Buffer.from(mods.Bzip2.compressFile(banana, null, 9));
    const damaged = (fn) => { const b = Buffer.from(good); fn(b); return b; };
    const errInputs = [
      ['garbage', Buffer.from('hello world, this is not bzip2'), false],
      ['short', Buffer.from('BZ'), false],
      ['level_0', damaged(b => { b[3] = 0x30; }), false],
      ['level_colon', damaged(b => { b[3] = 0x3a; }), false],
      ['block_magic_damaged', damaged(b => { b[5] ^= 0x01; }), false],
      ['block_crc_flipped', damaged(b => { b[10] ^= 0x80; }), false],
      ['stream_crc_flipped', damaged(b => { b[b.length - 2] ^= 0x10; }), false],
      ['randomised_bit', damaged(b => { b[14] |= 0x80; }), false],
      ['orig_pointer_huge', damaged(b => { b[14] = 0x7f; b[15] = 0xff; b[16] = 0xff; b[17] |= 0x80; }), false],
      ['payload_bit_flip', damaged(b => { b[24] ^= 0x04; }), false],
      ['truncated', good.slice(0, 20), false],
      ['second_member_bad_magic', Buffer.concat([good, Buffer.from('BZx9garbage')]), true],
      ['second_member_bad_level', Buffer.concat([good, Buffer.from('BZh0garbage')]), true],
      ['trailing_garbage_single_stream', Buffer.concat([good, Buffer.from('BZx9garbage')]), false],
    ];
    const errors = errInputs.map(([name, input, multi]) => {
      const rec = { name: name, input_hex: hex(input), multistream: multi };
      try {
        const out = mods.Bzip2.decompressFile(new Uint8Array(input), null, multi);
        rec.ok = true; rec.out_hex = hex(out);
      } catch (e) { rec.ok = false; rec.error_class = e.constructor.name; rec.message = e.message; rec.errorCode = e.errorCode === undefined ? null : e.errorCode; }
      return rec;
    });
    const levelErr = [0, 10, -1, 5.5].map(lv => {
      try { mods.Bzip2.compressFile(new Uint8Array(banana), null, lv); return { level: lv, ok: true }; }
      catch (e) { return { level: lv, ok: false, error_class: e.constructor.name, message: e.message }; }
    });
    const asStream = (buf, withSize) => { let pos = 0; const st = { readByte: function () { return pos < buf.length ? buf[pos++] : -1; } }; if (withSize) st.size = buf.length; return st; };
    const smallText = build({ kind: 'textgen', n: 3000, seed: 5 });
    const sizeless = [];
    [['empty', new Uint8Array(0)], ['banana', new Uint8Array(banana)], ['textgen_3000_s5', smallText]].forEach(([name, data]) => {
      [1, 9].forEach(level => {
        sizeless.push({ name: name, level: level, input_hex: name === 'textgen_3000_s5' ? null : hex(data),
                        no_size_hex: hex(mods.BWTC.compressFile(asStream(data, false), null, level)),
                        with_size_hex: hex(mods.BWTC.compressFile(asStream(data, true), null, level)),
                        array_hex: hex(mods.BWTC.compressFile(data, null, level)),
                        bzip2_no_size_hex: hex(mods.Bzip2.compressFile(asStream(data, false), null, level)) });
      });
    });
    const bwtcErrors = [Buffer.from('bwtx\x81\x09', 'binary'), Buffer.from('nope')].map(input => {
      try { mods.BWTC.decompressFile(new Uint8Array(input)); return { input_hex: hex(input), ok: true }; }
      catch (e) { return { input_hex: hex(input), ok: false, error_class: e.constructor.name, message: e.message }; }
    });
    const A = { kind: 'textgen', n: 150000, seed: 21 }, B = { kind: 'xorshift', n: 40000, seed: 77, mask: 15, add: 97 }, C = { kind: 'repeat', unit_hex: '616263', n: 1000 };
    const parts = [[A, 1], [B, 9], [C, 4]];
    const cat = Buffer.concat(parts.map(([r, lv]) => Buffer.from(mods.Bzip2.compressFile(build(r), null, lv))));
    const multiOut = mods.Bzip2.decompressFile(new Uint8Array(cat), null, true);
    const singleOut = mods.Bzip2.decompressFile(new Uint8Array(cat), null, false);
    const mixed = { parts: parts.map(([r, lv]) => ({ recipe: r, level: lv })), stream_len: cat.length, stream_sha256: sha256(cat),
                    multistream_out_len: multiOut.length, multistream_out_sha256: sha256(multiOut),
                    single_out_len: singleOut.length, single_out_sha256: sha256(singleOut) };
    fs.writeFileSync(path.join(HERE, 'golden_api.json'), JSON.stringify({ node: process.version, bzip2_decode_errors: errors, bzip2_level_errors: levelErr,
      bwtc_stream_input: sizeless, bwtc_decode_errors: bwtcErrors, mixed_level_multistream: mixed }, null, 1));
  } else {
    console.error('usage: make_golden.js small | big <job> | api');
    process.exit(2);
  }
}
main();
