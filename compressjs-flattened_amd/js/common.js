'use strict';
/* common.js — input/output coercion and error mapping shared by the Bzip2 / BWTC fronts.
 * Mirrors Util.coerceInputStream / Util.coerceOutputStream / BufferStream
 * (J/Bzip2_joined_.js:178-272): input = anything with readByte (a stream) or an indexable with
 * .length; output = nothing (returns a trimmed Uint8Array), an object with writeByte (bytes are pushed
 * there and that object is returned), a number (exact expected size) or a buffer filled exactly. */
var path = require('path');
var native = null;
function addon() {
  if (!native) { native = require(path.join(__dirname, 'cjs_napi.node')); }
  return native;
}

var EOF = -1;
function coerceInput(input) {
  if (input !== null && typeof input === 'object' && 'readByte' in input) {
    // drain the stream (the reference reads it byte by byte as well)
    var chunks = [], cur = new Uint8Array(65536), n = 0, total = 0, ch;
    while ((ch = input.readByte()) !== EOF && ch !== undefined && ch >= 0) {
      if (n === cur.length) { chunks.push(cur); total += n; cur = new Uint8Array(cur.length * 2); n = 0; }
      cur[n++] = ch;
    }
    var out = new Uint8Array(total + n), o = 0;
    chunks.forEach(function (c) { out.set(c, o); o += c.length; });
    out.set(cur.subarray(0, n), o);
    return { bytes: out, hasSize: ('size' in input && input.size >= 0) };
  }
  if (input instanceof Uint8Array) { return { bytes: input, hasSize: true }; }   // Buffer is a Uint8Array
  return { bytes: Uint8Array.from(input), hasSize: true };
}

function deliver(result, output) {
  if (!output) { return result; }
  if (typeof output === 'object' && 'writeByte' in output) {
    for (var i = 0; i < result.length; i++) { output.writeByte(result[i]); }
    if (output.flush) { output.flush(); }
    return output;
  }
  if (typeof output === 'number') {
    if (output !== result.length) { throw new TypeError('outputsize does not match decoded input'); }
    return result;
  }
  if (output.length !== result.length) { throw new TypeError('outputsize does not match decoded input'); }
  for (var j = 0; j < result.length; j++) { output[j] = result[j]; }
  return output;
}

module.exports = { addon: addon, coerceInput: coerceInput, deliver: deliver };
