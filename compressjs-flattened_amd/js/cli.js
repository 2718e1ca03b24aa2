#!/usr/bin/env node
'use strict';
/* cli.js — command line of the two fronts, with the switches of the reference's bin/compressjs for the algorithms in scope
 * (NPM/bin/compressjs:7-25 options, :31-58 checks and messages, :60-120 fd streams, :143-175 dispatch):
 *   cli.js -d|-z [-t bzip2|bwtc] [-1..-9] [-b <bits>] [infile] [outfile]
 * infile omitted: stdin; outfile omitted: stdout; neither -d nor -z: compress; default level 7 (:57).
 * The reference's default compressor (Lzp3) and its other -t values are not part of the MI355X core: -t must name bzip2
 * (alias bzip) or bwtc.  A file input knows its size, a pipe does not -- BWTC writes varint(0) then, as the reference does for a
 * stream without .size (:63-66; J/BWTC_joined_.js:529-543). */
var fs = require('fs');
var path = require('path');
var common = require(path.join(__dirname, 'common.js'));
var fronts = { bzip2: require(path.join(__dirname, 'Bzip2.js')), bwtc: require(path.join(__dirname, 'BWTC.js')) };

function fail(msg) { console.error(msg); process.exit(1); }

var argv = process.argv.slice(2), opt = { block: -1, files: [] }, level;
for (var i = 0; i < argv.length; i++) {
  var a = argv[i];
  if (a === '-d' || a === '--decompress') { opt.decompress = true; }
  else if (a === '-z' || a === '--compress') { opt.compress = true; }
  else if (a === '-b' || a === '--block') { opt.block = +argv[++i]; }
  else if (a === '-t') { opt.type = argv[++i]; }
  else if (/^-[1-9]$/.test(a)) {
    if (level) { fail("Can't specify both -" + level + ' and ' + a); }
    level = +a.slice(1);
  } else if (a === '-h' || a === '--help') {
    console.log('Usage: cli.js -d|-z [-t bzip2|bwtc] [-1..-9] [-b <bits>] [infile] [outfile]\n' +
                '  If <infile> is omitted, reads from stdin.\n  If <outfile> is omitted, writes to stdout.');
    process.exit(0);
  } else if (a[0] === '-' && a.length > 1) { fail('Unknown option: ' + a); }
  else { opt.files.push(a); }
}
if (!opt.decompress) { opt.compress = true; }
if (opt.decompress && opt.compress) { fail('Must specify either -d or -z.'); }
if (opt.compress && opt.block >= 0) { fail('--block can only be used with decompression'); }
if (level && opt.decompress) { fail('Compression level has no effect when decompressing.'); }
if (!level) { level = 7; }
var type = String(opt.type || '').toLowerCase();
if (type === 'bzip') { type = 'bzip2'; }
if (!fronts[type]) {
  fail(opt.type ? 'Unknown compressor: ' + opt.type + ' (this build has bzip2 and bwtc)' : 'Select the compressor with -t bzip2 or -t bwtc');
}

function readAll(fd) {
  var st = fs.fstatSync(fd);
  if (st.isFile() && st.size > 0) {
    var buf = Buffer.allocUnsafe(st.size), got = 0, n;
    while (got < st.size && (n = fs.readSync(fd, buf, got, st.size - got, null)) > 0) { got += n; }
    return { bytes: buf.slice(0, got), hasSize: true };
  }
  var chunks = [], chunk = Buffer.allocUnsafe(1 << 20), m, total = 0;
  for (;;) {
    try { m = fs.readSync(fd, chunk, 0, chunk.length, null); } catch (e) { if (e.code === 'EAGAIN') { continue; } if (e.code === 'EOF') { break; } throw e; }
    if (m <= 0) { break; }
    chunks.push(Buffer.from(chunk.slice(0, m))); total += m;
  }
  return { bytes: Buffer.concat(chunks, total), hasSize: false };
}
function writeAll(fd, bytes) {
  var buf = Buffer.from(bytes.buffer, bytes.byteOffset, bytes.length), off = 0;
  while (off < buf.length) { off += fs.writeSync(fd, buf, off, Math.min(buf.length - off, 1 << 24)); }
}

var inFd = opt.files.length > 0 ? fs.openSync(opt.files[0], 'r') : 0;
var outFd = opt.files.length > 1 ? fs.openSync(opt.files[1], 'w') : 1;
var input = readAll(inFd), result;
try {
  if (opt.decompress) {
    if (opt.block >= 0) {
      if (type !== 'bzip2') { fail('--block needs -t bzip2'); }
      result = fronts.bzip2.decompressBlock(input.bytes, opt.block);
    } else { result = fronts[type].decompressFile(input.bytes); }
  } else if (type === 'bwtc' && !input.hasSize) {
    result = common.addon().bwtcCompress(input.bytes, level, 1 /* CJS_FLAG_SIZE_UNKNOWN */);
  } else { result = fronts[type].compressFile(input.bytes, null, level); }
} catch (e) { fail(String(e && e.message ? e.message : e)); }
writeAll(outFd, result);
if (inFd !== 0) { fs.closeSync(inFd); }
if (outFd !== 1) { fs.closeSync(outFd); }
