'use strict';
/* Bzip2.js — drop-in front for the reference's `Bzip2` object (J/Bzip2_joined_.js:2198-2253):
 * same method names, argument meaning and error behaviour; the per-block pipeline runs in
 * libcjs_hip.so on an MI355X through the N-API addon.  No JavaScript fallback. */
var common = require('./common.js');

var Err = { OK: 0, LAST_BLOCK: -1, NOT_BZIP_DATA: -2, UNEXPECTED_INPUT_EOF: -3, UNEXPECTED_OUTPUT_EOF: -4,
            DATA_ERROR: -5, OUT_OF_MEMORY: -6, OBSOLETE_INPUT: -7, END_OF_BLOCK: -8 };
var Messages = {};
Messages[Err.NOT_BZIP_DATA] = 'Not bzip data';
Messages[Err.DATA_ERROR] = 'Data error';
Messages[Err.OUT_OF_MEMORY] = 'Out of memory';
Messages[Err.OBSOLETE_INPUT] = 'Obsolete (pre 0.9.5) bzip format not supported.';

function rethrow(e) {
  if (e && typeof e.cjsCode === 'number') {
    var code = e.cjsCode;
    if (code === -20) { throw new Error('Invalid block size multiplier'); }      // J/Bzip2_joined_.js:2208
    if (Messages[code]) {                                                       // _throw(status, optDetail) :1385-1391
      var msg = Messages[code];
      if (e.cjsDetail) { msg += ': ' + e.cjsDetail; }
      var t = new TypeError(msg); t.errorCode = code; throw t;
    }
    var g = new Error(e.message); g.errorCode = code; throw g;
  }
  throw e;
}

var Bzip2 = Object.create(null);
Bzip2.compressFile = function (inStream, outStream, props) {
  var level = 9;
  if (typeof props === 'number') { level = props; }
  if (level < 1 || level > 9) { throw new Error('Invalid block size multiplier'); }
  var input = common.coerceInput(inStream);
  var result;
  try { result = common.addon().bzip2Compress(input.bytes, level); } catch (e) { rethrow(e); }
  return common.deliver(result, outStream);
};
Bzip2.decompressFile = function (inStream, outStream, multistream) {
  var input = common.coerceInput(inStream);
  var result;
  try { result = common.addon().bzip2Decompress(input.bytes, multistream ? 1 : 0); } catch (e) { rethrow(e); }
  return common.deliver(result, outStream);
};
// Bunzip.decodeBlock (J/Bzip2_joined_.js:1797-1818): the block whose magic starts at bit `pos`
Bzip2.decompressBlock = function (inStream, pos, outStream) {
  var input = common.coerceInput(inStream);
  var result;
  try { result = common.addon().bzip2DecompressBlock(input.bytes, pos); } catch (e) { rethrow(e); }
  return common.deliver(result, outStream);
};
// Bunzip.table (:1823-1863): callback(position in bits, uncompressed size in bytes) once per block
Bzip2.table = function (inStream, callback, multistream) {
  var input = common.coerceInput(inStream);
  var t;
  try { t = common.addon().bzip2Table(input.bytes, multistream ? 1 : 0); } catch (e) { rethrow(e); }
  for (var i = 0; i < t.length; i += 2) { callback(t[i], t[i + 1]); }
};
Bzip2.Err = Err;
module.exports = Bzip2;
